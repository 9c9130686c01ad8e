# usage: bash tools/gpu_job_r2_w.sh <tag> -- parity tests, then c2 bench with k_sync register budgets for 4 and 3 workgroups per CU
TAG=${1:-x}
mkdir -p gpurun_out/r2_$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_$TAG/pytest.log 2>&1; rc=$?; echo pytest exit=$rc; tail -3 gpurun_out/r2_$TAG/pytest.log
[ $rc -eq 0 ] || exit $rc
for W in 4 3 4 3; do
OFDM_SYNC_W=$W timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-packets 0 > gpurun_out/r2_$TAG/bench_w$W.json 2> gpurun_out/r2_$TAG/bench_w$W.err; echo W=$W bench exit=$?
python tools/show_bench.py gpurun_out/r2_$TAG/bench_w$W.json 2>/dev/null || tail -c 800 gpurun_out/r2_$TAG/bench_w$W.err
done
