mkdir -p gpurun_out/r2_es
for c in c3 c5; do
for es in 0 768 1024 1280 1536 1792 2048; do
OFDM_EXACT_SMALL=$es timeout -k 10 300 python bench.py --config $c --cpu-packets 0 --steps 4 --warmup 1 --no-pipeline > gpurun_out/r2_es/b_${c}_$es.json 2> gpurun_out/r2_es/b_${c}_$es.err
echo -n "$c es=$es  "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2_es/b_${c}_$es.json").read().strip().splitlines()[-1]); print("exact %.3f  sync %.3f  step %.3f" % (d["kernels_ms_per_step"]["k_sync_exact"], d["kernels_ms_per_step"]["k_sync"], d["ms_per_step"]))
except Exception as e:
    print("failed", e)
PY
done
done
