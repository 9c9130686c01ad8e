# usage: bash tools/gpu_job_r3.sh <tag> [pytest args...]   -- GPU tests (default: the whole -m gpu suite), then a short C2 bench
TAG=${1:-x}; shift
ARGS=${@:-tests -m gpu -x -q}
mkdir -p gpurun_out/r3_$TAG
timeout -k 10 900 python -m pytest $ARGS > gpurun_out/r3_$TAG/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -25 gpurun_out/r3_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r3_$TAG/bench_c2.json 2> gpurun_out/r3_$TAG/bench_c2.err; echo bench exit=$?
python tools/show_bench.py gpurun_out/r3_$TAG/bench_c2.json 2>/dev/null || tail -c 1500 gpurun_out/r3_$TAG/bench_c2.err
