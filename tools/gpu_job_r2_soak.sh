# usage: bash tools/gpu_job_r2_soak.sh <tag> [seconds-per-soak] [seed]  -- the randomised soaks of tests/soak on the GPU box
TAG=${1:-x}; SEC=${2:-150}; SEED=${3:-2}
O=gpurun_out/soak_$TAG; mkdir -p $O
for s in fuzz_parity fuzz_stream fuzz_reuse fuzz_sense; do
  timeout -k 10 $((SEC + 240)) python tests/soak/$s.py $SEC $SEED > $O/$s.log 2>&1; echo "$s exit=$?"; tail -1 $O/$s.log
  grep -c MISMATCH $O/$s.log > /dev/null && grep MISMATCH $O/$s.log | cut -c1-300 | sed -n 1,5p
done
timeout -k 10 300 python tests/soak/edge_inputs.py > $O/edge_inputs.log 2>&1; echo "edge exit=$?"; tail -2 $O/edge_inputs.log
true
