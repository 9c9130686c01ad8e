# needs the diagnostic library: make -C ofdm_uhd_amd/csrc diag (built before the gpurun call)
export OFDM_HIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/ofdm_uhd_amd/csrc/libofdm_hip_diag.so
for a in 0 1 2 3 4 7; do
  OFDM_ABLATE=$a timeout -k 10 200 python bench.py --packets 16384 --steps 3 --warmup 1 --cpu-packets 0 > gpurun_out/ab_$a.log 2>&1
  echo -n "ablate=$a  "; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$a.log").read().strip().splitlines()[-1]); print("k_sync %.3f ms" % d["kernels_ms_per_step"]["k_sync"])
except Exception as e:
    print("failed", e, open("gpurun_out/ab_$a.log").read()[-300:])
PY
done
