# usage: bash tools/gpu_job_r2_libs2.sh <tag> <lib>... -- c2 bench of each build in turn, pipelined and sequential (twice)
TAG=${1:-x}; shift
mkdir -p gpurun_out/r2_$TAG
for rep in 1 2; do
for L in "$@"; do
for MODE in "" "--no-pipeline"; do
OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/$L timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-packets 0 $MODE > gpurun_out/r2_$TAG/bench_$L$MODE.json 2> gpurun_out/r2_$TAG/bench_$L$MODE.err; echo "$L $MODE exit=$?"
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$L$MODE.json 2>/dev/null | head -1 || tail -c 800 gpurun_out/r2_$TAG/bench_$L$MODE.err
done
done
done
