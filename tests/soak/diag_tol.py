"""Worst float deviations GPU vs oracle per RX tap for one parity case: python tests/soak/diag_tol.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import loopback_stream, make_cfg, make_payloads
from ofdm_uhd_amd import _abi, engine
from oracle import oracle as orc
maps = [b["carrier_map"] for b in json.load(open(os.path.join(ROOT, "tests/golden/sense_blocks.json")))["blocks"]]
for which in (0, 17, 36, None):
    carriers = maps[which][:50] if which is not None else None
    cfg = make_cfg("qpsk", 512, 200, 128, carriers=carriers)
    eng = engine.Engine(cfg=cfg)
    pay = make_payloads(5, 700, seed=which or 0)
    x = loopback_stream(orc, cfg, pay, snr_db=30.0)
    taps = (_abi.TAP_RX_FFT, _abi.TAP_RX_ACQ, _abi.TAP_RX_SINK)
    mask = 0
    for t in taps: mask |= 1 << t
    ro = orc.rx(cfg, x, mask)
    eng.set_taps(*taps); eng.rx(x)
    for t in taps:
        a, b = ro.tap(t), eng.tap(t)
        r = np.abs(a - b) / np.maximum(1.0, np.abs(a))
        i = np.unravel_index(np.argmax(r), r.shape)
        print(which, "tap", t, "max rel %.3g at %s ref %s gpu %s |ref|max %.3g  frac>1e-5 %.2g" % (r.max(), i, a[i], b[i], np.abs(a).max(), (r > 1e-5).mean()))
    eng.close()
