# usage: bash tools/gpu_job_r3_soakall.sh <tag> <seed>  -- the round's soak: every randomised soak + the edge inputs
TAG=${1:-x}; SEED=${2:-2}
O=gpurun_out/soak3_$TAG; mkdir -p $O
run() { timeout -k 10 $(($2 + 200)) python tests/soak/$1.py $2 $SEED > $O/$1.log 2>&1; echo "$1 exit=$?"; tail -n 1 $O/$1.log; grep MISMATCH $O/$1.log | cut -c1-300 | sed -n 1,4p; }
run fuzz_parity 300
run fuzz_reuse 250
run fuzz_stream 150
run fuzz_sense 80
timeout -k 10 300 python tests/soak/edge_inputs.py > $O/edge_inputs.log 2>&1; echo "edge exit=$?"; tail -n 2 $O/edge_inputs.log
true
