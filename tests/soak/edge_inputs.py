"""Hostile inputs must end in a result or a clean error, never a hang: python tests/soak/edge_inputs.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_cfg
from ofdm_uhd_amd import engine, config
from oracle import oracle as orc

rng = np.random.default_rng(0)
eng = engine.Engine(cfg=make_cfg("qpsk"))
cfg = eng.cfg
def run(name, x, compare=True):
    x = np.ascontiguousarray(x, np.complex64)
    try:
        got = eng.rx(x)
        st = dict(eng.last_stats)
        msg = "%d packets, stats %s" % (len(got), {k: st[k] for k in ("peaks", "frames", "packets", "overflow")})
        if compare:
            ro = orc.rx(cfg, x)
            msg += " | oracle packets equal: %s, peaks %s" % (got == ro.packets, ro.stats["peaks"] == st["peaks"])
    except Exception as e:          # EngineError / ValueError are clean outcomes
        msg = "%s: %s" % (type(e).__name__, str(e)[:100])
    print("%-34s %s" % (name, msg), flush=True)

run("empty", np.zeros(0))
run("1 sample", np.ones(1))
run("all zeros 100k", np.zeros(100000))
run("N-1 samples", np.zeros(511) + 0.1)
run("constant 1.0 x 300k", np.ones(300000))
run("pure tone 300k", np.exp(2j * np.pi * 0.01 * np.arange(300000)))
run("period-256 sequence 300k", np.tile(rng.standard_normal(256) + 1j * rng.standard_normal(256), 1200))
run("white noise 1M", (rng.standard_normal(1000000) + 1j * rng.standard_normal(1000000)) * 0.01)
x = (rng.standard_normal(200000) + 1j * rng.standard_normal(200000)).astype(np.complex64) * 0.01
x[1000] = np.nan
run("noise with one NaN", x, compare=False)
x[1000] = np.inf
run("noise with one Inf", x, compare=False)
run("huge amplitude 1e30", (rng.standard_normal(200000) + 1j * rng.standard_normal(200000)) * 1e30, compare=False)
run("denormal amplitude 1e-42", (rng.standard_normal(200000) + 1j * rng.standard_normal(200000)) * 1e-42)
# sensing on hostile input
sc = config.make_sense_cfg(256, 1, 3, 2, 1)
for name, v in (("sense zeros", np.zeros(50000)), ("sense NaN", np.full(50000, np.nan)), ("sense 1e30", np.ones(50000) * 1e30)):
    r = eng.sense(sc, v.astype(np.complex64))
    print("%-34s %d msgs, hex %s" % (name, len(r["msgs"]), r["hex"][:1]))
print("edge inputs done")
