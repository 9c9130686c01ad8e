# usage: bash tools/gpu_job_r3_np.sh <tag> [config]  -- pipelined and sequential bench of one build (+ the four-round-trip receiver)
TAG=${1:-x}; CFG=${2:-c2}
mkdir -p gpurun_out/r3_$TAG
for mode in pipe seq seq4; do
  ARGS=""; [ $mode != pipe ] && ARGS="--no-pipeline"
  [ $mode = seq4 ] && export OFDM_RX_SYNCS=1
  timeout -k 10 300 python bench.py --config $CFG --steps 8 --warmup 2 --cpu-packets 0 $ARGS > gpurun_out/r3_$TAG/bench_${CFG}_$mode.json 2> gpurun_out/r3_$TAG/bench_${CFG}_$mode.err; echo "$mode exit=$?"
  python tools/show_bench.py gpurun_out/r3_$TAG/bench_${CFG}_$mode.json 2>/dev/null | head -n 1
  unset OFDM_RX_SYNCS
done
