#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) of tools/gpu_job_prof.sh."""
import collections, csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/trace/*/*_kernel_stats.csv"):
    print("== kernel stats")
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-34s calls %3s avg %10.1f us  %5.1f%%" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for sub in ("pmc1", "pmc2", "pmc3", "pmc4"):
    fs = glob.glob(d + "/" + sub + "/*/*_counter_collection.csv")
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.Counter()
    first = rows[0]["Counter_Name"]
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first:
            seen[k] += 1
    print("==", sub)
    for k in agg:
        if not any(t in k for t in ("k_sync", "k_rx_demod", "k_tx_mod", "k_deframe_write", "k_frame_pack", "k_peak", "k_chan_filter", "k_sense")):
            continue
        n = seen[k]
        print("  %-30s x%d " % (k, n) + " ".join("%s=%.4g" % (c.replace("SQ_", ""), v / n) for c, v in sorted(agg[k].items())))
