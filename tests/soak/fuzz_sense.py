"""Randomised soak of the spectrum sensor: engine vs oracle.  python tests/soak/fuzz_sense.py [seconds] [seed]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_cfg
from ofdm_uhd_amd import config, engine
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = engine.Engine(cfg=make_cfg("qpsk"))
t_end = time.time() + budget
ncase = nbad = 0
while time.time() < t_end:
    S = int(rng.choice([64, 128, 256, 256, 512, 1024, 2048, 4096]))
    tune, dwell = int(rng.integers(0, 6)), int(rng.integers(1, 40))
    avg, skip = int(rng.integers(1, 12)), int(rng.integers(0, 3))
    thr = float(rng.choice([1e-4, 1e-3, 0.2]))
    win = None if rng.random() < 0.7 else (0.5 + rng.random(S)).tolist()
    sc = config.make_sense_cfg(S, tune, dwell, avg, skip, thr, win)
    nmsg = int(rng.integers(0, 3 * (avg + skip) + 2))
    n = (tune + dwell) * S * nmsg + int(rng.integers(0, S))
    n = min(n, 6_000_000)
    floor = float(rng.choice([1e-7, 1e-6, 1e-5]))
    iq = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(floor / (0.52 * S))).astype(np.complex64)
    t = np.arange(n)
    for _ in range(int(rng.integers(0, 4))):
        f, a = rng.random(), 10 ** rng.uniform(-3, -0.5)
        iq += (a / (0.36 * S) * np.exp(2j * np.pi * f * t)).astype(np.complex64)
    desc = dict(S=S, tune=tune, dwell=dwell, avg=avg, skip=skip, thr=thr, n=n, window=win is not None)
    ncase += 1
    try:
        g, o = eng.sense(sc, iq), orc.sense(sc, iq)
        assert g["msgs"].shape == o["msgs"].shape and len(g["hex"]) == len(o["hex"]), "shapes"
        assert np.array_equal(g["msgs"], o["msgs"]), "msgs"
        if len(o["hex"]):
            assert np.array_equal(g["mean"], o["mean"]) and np.array_equal(g["bits"], o["bits"]) and g["hex"] == o["hex"], "decisions"
            d = orc.sense_decide(sc, g["msgs"])            # the tail on the GPU's own messages: exact
            assert np.array_equal(d["mean"], g["mean"]) and np.array_equal(d["bits"], g["bits"]) and d["hex"] == g["hex"], "tail"
    except AssertionError as e:
        nbad += 1
        print("MISMATCH [%s]" % e, json.dumps(desc), flush=True)
print("fuzz_sense: %d cases, %d mismatches, seed %d" % (ncase, nbad, seed))
