#!/usr/bin/env python3
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("sym/s %.3e  ms/step %.3f  crc %.6f  roofline (path) %.4f  dominant %s %.4f" % (d["value"], d["ms_per_step"], d["crc_pass_rate"], r["frac"], r["kernel"], r.get("kernel_frac", r["frac"])))
print("  ".join("%s=%.3f" % (k, v) for k, v in d["kernels_ms_per_step"].items()))
if d.get("cpu_baseline"):
    print(d["cpu_baseline"])
