# usage: bash tools/gpu_job_r2_prio.sh <tag> <lib>... -- c2 pipelined bench of each build with and without stream priorities
TAG=${1:-x}; shift
mkdir -p gpurun_out/r2_$TAG
for rep in 1 2; do
for L in "$@"; do
for PR in 1 0; do
OFDM_STREAM_PRIO=$PR OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/$L timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_$L.$PR.json 2> gpurun_out/r2_$TAG/bench_$L.$PR.err; echo "$L prio=$PR exit=$?"
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$L.$PR.json 2>/dev/null | head -1 || tail -c 800 gpurun_out/r2_$TAG/bench_$L.$PR.err
done
done
done
