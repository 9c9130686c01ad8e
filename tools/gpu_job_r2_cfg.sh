# usage: bash tools/gpu_job_r2_cfg.sh <tag> <config> <lib>... -- bench of one config with each build in turn (twice)
TAG=${1:-x}; CFG=$2; shift 2
mkdir -p gpurun_out/r2_$TAG
for rep in 1 2; do
for L in "$@"; do
OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/$L timeout -k 10 300 python bench.py --config $CFG --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_$L.json 2> gpurun_out/r2_$TAG/bench_$L.err; echo $L bench exit=$?
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$L.json 2>/dev/null || tail -c 800 gpurun_out/r2_$TAG/bench_$L.err
done
done
