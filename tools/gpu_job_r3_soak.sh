# usage: bash tools/gpu_job_r3_soak.sh <tag> [seconds-per-soak] [seed] [soaks...]  -- randomised soaks of tests/soak on the GPU box
TAG=${1:-x}; SEC=${2:-150}; SEED=${3:-2}; shift 3
SOAKS=${@:-fuzz_parity fuzz_stream fuzz_reuse fuzz_sense}
O=gpurun_out/soak3_$TAG; mkdir -p $O
for s in $SOAKS; do
  timeout -k 10 $((SEC + 240)) python tests/soak/$s.py $SEC $SEED > $O/$s.log 2>&1; echo "$s exit=$?"; tail -n 1 $O/$s.log
  grep MISMATCH $O/$s.log | cut -c1-400 | sed -n 1,6p
done
true
