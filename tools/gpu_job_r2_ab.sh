# usage: bash tools/gpu_job_r2_ab.sh <tag> <libA> <libB> -- parity subset with the default library, then alternating c2 benches of two builds
TAG=${1:-x}; A=$2; B=$3
mkdir -p gpurun_out/r2_$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2_$TAG/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -3 gpurun_out/r2_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
for L in $A $B $A $B; do
OFDM_HIP_LIB=$PWD/ofdm_uhd_amd/csrc/$L timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_$L.json 2> gpurun_out/r2_$TAG/bench_$L.err; echo $L bench exit=$?
python tools/show_bench.py gpurun_out/r2_$TAG/bench_$L.json 2>/dev/null || tail -c 800 gpurun_out/r2_$TAG/bench_$L.err
done
