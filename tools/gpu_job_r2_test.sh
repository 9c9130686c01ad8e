# usage: bash tools/gpu_job_r2_test.sh <tag>   -- GPU parity suite, then a short c2 bench (no CPU leg)
TAG=${1:-x}
mkdir -p gpurun_out/r2_$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_$TAG/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -5 gpurun_out/r2_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_c2.json 2> gpurun_out/r2_$TAG/bench_c2.err; echo bench exit=$?
python tools/show_bench.py gpurun_out/r2_$TAG/bench_c2.json 2>/dev/null || tail -c 1500 gpurun_out/r2_$TAG/bench_c2.json
