# usage: bash tools/gpu_job_r3_stamps.sh <tag> [config]  -- product bench, then the stamps build's per-phase profile of k_sync / the fused front end
TAG=${1:-x}; CFG=${2:-c2}
mkdir -p gpurun_out/r3_$TAG
timeout -k 10 300 python bench.py --config $CFG --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r3_$TAG/bench_$CFG.json 2> gpurun_out/r3_$TAG/bench_$CFG.err; echo bench exit=$?
python tools/show_bench.py gpurun_out/r3_$TAG/bench_$CFG.json 2>/dev/null || tail -c 1500 gpurun_out/r3_$TAG/bench_$CFG.err
OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_stamps.so timeout -k 10 300 python bench.py --config $CFG --steps 1 --warmup 1 --cpu-packets 0 --no-pipeline > gpurun_out/r3_$TAG/stamps.json 2> gpurun_out/r3_$TAG/stamps.err; tail -n 12 gpurun_out/r3_$TAG/stamps.err
