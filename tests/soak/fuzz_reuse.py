"""One engine, many calls of very different sizes and modes (taps on/off, fused sensing on/off, carrier map
switched): workspaces grow, shrink in use and are reused -- every call must still equal the oracle.
python tests/soak/fuzz_reuse.py [seconds] [seed]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_cfg, make_payloads
from ofdm_uhd_amd import _abi, engine, config
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
maps = [b["carrier_map"][:50] for b in json.load(open(os.path.join(ROOT, "tests/golden/sense_blocks.json")))["blocks"]]
t_end = time.time() + budget
ncase = nbad = 0
for gi, (mod, N, occ, CP) in enumerate((("qpsk", 512, 200, 128), ("qam16", 1024, 600, 256))):
    if os.environ.get("FUZZ_GEOM") not in (None, str(gi)):
        continue
    rng = np.random.default_rng([seed, gi])   # (per geometry: the second one's cases do not depend on how many the first ran)
    t_beat = time.time() + 30
    cfg = make_cfg(mod, N, occ, CP)
    eng = engine.Engine(cfg=cfg)
    carriers = ""
    sense_on = False
    t_geo = time.time() + budget / 2
    while time.time() < t_geo:
        ncase += 1
        if time.time() > t_beat:
            print("... %d calls, %d mismatches" % (ncase, nbad), flush=True)
            t_beat = time.time() + 30
        if N == 512 and rng.random() < 0.15:
            carriers = "" if rng.random() < 0.4 else maps[int(rng.integers(0, len(maps)))]
            eng.set_carrier_map(carriers)
            cfg = make_cfg(mod, N, occ, CP, carriers=carriers or None)
        if rng.random() < 0.2:
            sense_on = not sense_on
            eng.set_rx_sense(config.make_sense_cfg(256, 1, 5, 2, 1) if sense_on else None)
        npkt = int(rng.choice([0, 1, 2, 5, 20, 60]))
        sizes = rng.integers(0, 2500, npkt)
        pay = make_payloads(npkt, sizes, seed=int(rng.integers(0, 1 << 30)))
        lead, tail = int(rng.integers(0, 2000)), int(rng.integers(0, 3000))
        desc = dict(mod=mod, npkt=npkt, lead=lead, tail=tail, carriers=carriers, sense=sense_on)
        try:
            iq_o = orc.tx(cfg, pay)
            iq_g = eng.tx(pay)
            assert np.array_equal(iq_g, iq_o), "tx"
            x = np.concatenate([np.zeros(lead, np.complex64), iq_o, np.zeros(tail, np.complex64)])
            core = iq_o if len(iq_o) else np.ones(1, np.complex64)
            orc.channel(x, sigma=float(np.sqrt(np.mean(np.abs(core) ** 2) / 10 ** float(rng.choice([2.0, 3.0, 4.0])))),
                        cfo=float(rng.choice([0.0, 0.05, -0.3])) * 2 * np.pi / N, seed=int(rng.integers(0, 1 << 30)))
            taps = () if rng.random() < 0.5 else (_abi.TAP_RX_METRIC, _abi.TAP_RX_FFT, _abi.TAP_RX_PACKETS)
            eng.set_taps(*taps)
            ro = orc.rx(cfg, x)
            pk = eng.rx(x)
            pg, po = eng.tap(_abi.TAP_RX_PEAKS).tolist(), ro.tap(_abi.TAP_RX_PEAKS).tolist()
            if pg != po:
                d = sorted(set(pg) ^ set(po))[:6]
                print("peaks differ: gpu %d oracle %d; symmetric difference (first) %s" % (len(pg), len(po), d))
                u = orc.rx(cfg, x, 1 << _abi.TAP_RX_METRIC).tap(_abi.TAP_RX_METRIC)
                for q0 in d[:3]:
                    print("   u around %d (in gpu: %s, in oracle: %s): %s" % (q0, q0 in pg, q0 in po, np.array2string(u[max(0, q0 - 40):q0 + 8], precision=6, max_line_width=200)))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "reuse_fail_%d.npy" % ncase), x)
            for k in ("symbols", "peaks", "frames", "headers_ok", "packets", "chained_frames"):
                assert eng.last_stats[k] == ro.stats[k], k
            assert pk == ro.packets, "packets"
            if sense_on:
                r = eng.rx_sense_result(len(x))
                o = orc.sense(config.make_sense_cfg(256, 1, 5, 2, 1), x)
                assert r["msgs"].shape == o["msgs"].shape, "sense shape"
                assert np.array_equal(r["msgs"], o["msgs"]), "sense msgs"
        except (AssertionError, engine.EngineError, ValueError) as e:
            nbad += 1
            if os.environ.get("FUZZ_STOP"):
                t_end = t_geo = 0
            print("MISMATCH [%s]" % (e if isinstance(e, AssertionError) else "error " + str(e)[:90]), json.dumps(desc), flush=True)
    eng.close()
print("fuzz_reuse: %d calls, %d mismatches, seed %d" % (ncase, nbad, seed))
