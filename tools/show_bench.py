#!/usr/bin/env python3
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("sym/s %.3e  ms/step %.3f  crc %.6f  roofline %s %.4f path %.4f" % (d["value"], d["ms_per_step"], d["crc_pass_rate"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["path_frac"]))
print("  ".join("%s=%.3f" % (k, v) for k, v in d["kernels_ms_per_step"].items()))
if d.get("cpu_baseline"):
    print(d["cpu_baseline"])
