"""Standalone timing of the sensing kernel (k_sense + decision tail) on a resident IQ buffer:
python tools/bench_sense.py [nsamples_millions]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ofdm_uhd_amd import config, engine, options

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda", 0)
x = (torch.randn(2 * n, device=dev) * 1e-3)
eng = engine.Engine(cfg=config.make_cfg(options.default_options(modulation="qpsk"), device_ptrs=True))
for S in (64, 256, 512, 1024, 2048, 4096):
    tune = max(0, int(round(1e-3 * 6.25e6 / S)))
    dwell = max(1, int(round(10e-3 * 6.25e6 / S)))
    sc = config.make_sense_cfg(S, tune, dwell, 10, 1)
    eng.prof_enable(True)
    for it in range(3):
        if it == 1:
            eng.prof_reset()
        r = eng.sense(sc, x.data_ptr(), n)
    ms, cnt = eng.prof()["k_sense"]
    nm = len(r["msgs"])
    used = nm * dwell * S * 8.0
    print("S=%4d tune=%3d dwell=%3d msgs=%6d dec=%5d  %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
        S, tune, dwell, nm, len(r["hex"]), ms / cnt, used / (ms / cnt * 1e-3) / 1e9, used / (ms / cnt * 1e-3) / 8e12 * 100))
