#!/usr/bin/env python3
"""profiles/pmc_traffic_<config>.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_job_prof.sh.

    python tools/make_pmc_traffic.py gpurun_out/prof_<tag> <config> <symbols per launch> <tag for `source`>

Counters are KiB per dispatch; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide
coalesced reads on gfx950.  The file carries a hash of the kernel sources it was measured on: bench.py only quotes
it while the sources are unchanged."""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ofdm_uhd_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


KERNELS = {"k_frame_pack": "k_frame_pack", "k_tx_mod": "k_tx_mod", "k_chan_filter": "k_chan_filter", "k_sync<": "k_sync",
           "k_sync_exact": "k_sync_exact", "k_rx_demod": "k_rx_demod", "k_deframe_write": "k_deframe", "k_sense<": "k_sense"}


def per_kernel(path, counter):
    # (a scratch directory may hold the files of several runs: the newest one with counters in it)
    fs = sorted(glob.glob(path + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    fs = [f for f in fs if os.path.getsize(f) > 1000]
    tot, steps = collections.defaultdict(float), 0
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] != counter:
            continue
        if "k_rx_demod" in r["Kernel_Name"]:
            steps += 1  # (one launch per step; k_sync_exact has two: per-step sums, not per-launch averages)
        for pat, name in KERNELS.items():
            if pat in r["Kernel_Name"]:
                tot[name] += float(r["Counter_Value"])
    return {k: tot[k] / steps for k in tot}


def main():
    d, cfg, nsym, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rd, wr = per_kernel(d + "/pmc3", "FETCH_SIZE"), per_kernel(d + "/pmc4", "WRITE_SIZE")
    out = {"source": tag, "sources_sha": sources_sha(), "config": cfg, "symbols_per_launch": nsym,
           "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_job_prof.sh); "
                  "KiB per step; FETCH_SIZE doubled (MI355X_MICROARCH.md, gfx950 wide reads)",
           "bytes_per_symbol": {k: {"read": 2.0 * rd.get(k, 0.0) * 1024.0 / nsym, "write": wr.get(k, 0.0) * 1024.0 / nsym}
                                for k in sorted(set(rd) | set(wr))}}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % cfg), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
