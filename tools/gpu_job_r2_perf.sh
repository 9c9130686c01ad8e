# usage: bash tools/gpu_job_r2_perf.sh <tag> [config] -- quick parity subset, product bench, then the stamps build's k_sync phase profile
TAG=${1:-x}; CFG=${2:-c2}
mkdir -p gpurun_out/r2_$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2_$TAG/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -3 gpurun_out/r2_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config $CFG --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_$CFG.json 2> gpurun_out/r2_$TAG/bench_$CFG.err; echo bench exit=$?
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$CFG.json 2>/dev/null || tail -c 1500 gpurun_out/r2_$TAG/bench_$CFG.err
if [ -f ofdm_uhd_amd/csrc/libofdm_hip_stamps.so ]; then
OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/libofdm_hip_stamps.so timeout -k 10 300 python bench.py --config $CFG --steps 1 --warmup 1 --cpu-packets 0 > gpurun_out/r2_$TAG/stamps.json 2> gpurun_out/r2_$TAG/stamps.err; tail -15 gpurun_out/r2_$TAG/stamps.err
fi
