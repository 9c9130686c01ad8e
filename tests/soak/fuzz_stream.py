"""Randomised soak of ofdm_demod.feed(): random captures cut into random chunks must give the one-shot packets.
python tests/soak/fuzz_stream.py [seconds] [seed]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_cfg, make_payloads
from ofdm_uhd_amd import ofdm, options
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
ncase = nbad = 0
GEOMS = [("qpsk", 512, 200, 128), ("bpsk", 512, 200, 128), ("qam16", 1024, 600, 256), ("8psk", 256, 120, 64),
         ("qam64", 2048, 1200, 512), ("qpsk", 128, 64, 32)]
while time.time() < t_end:
    mod, N, occ, CP = GEOMS[int(rng.integers(0, len(GEOMS)))]
    cfg = make_cfg(mod, N, occ, CP)
    opt = options.default_options(modulation=mod, fft_length=N, occupied_tones=occ, cp_length=CP)
    npkt = int(rng.integers(5, 60))
    sizes = rng.integers(0, 3000, npkt)
    if rng.random() < 0.3:
        sizes[int(rng.integers(0, npkt))] = 4091
    pay = make_payloads(npkt, sizes, seed=int(rng.integers(0, 1 << 30)))
    parts, k = [np.zeros(int(rng.integers(0, 5000)), np.complex64)], 0
    while k < npkt:
        n = int(rng.integers(1, 10))
        parts.append(orc.tx(cfg, pay[k:k + n]))
        parts.append(np.zeros(int(rng.choice([0, 50, 3000, 40000, 700000 // max(1, npkt)])), np.complex64))
        k += n
    iq = np.concatenate(parts)
    core = parts[1]
    snr = float(rng.choice([15.0, 25.0, 30.0, 40.0, 60.0, 80.0]))
    cfo = float(rng.choice([0.0, 0.05, -0.3, 1.2]))
    orc.channel(iq, sigma=float(np.sqrt(np.mean(np.abs(core) ** 2) / 10 ** (snr / 10))), cfo=cfo * 2 * np.pi / N,
                seed=int(rng.integers(0, 1 << 30)))
    if rng.random() < 0.2:
        # a carrier over a stretch (or all) of the capture
        amp = float(np.sqrt(np.mean(np.abs(core) ** 2)) * 10 ** rng.uniform(-1.5, 0.7))
        a, b = (0, len(iq)) if rng.random() < 0.3 else sorted(rng.integers(0, len(iq), 2).tolist())
        iq[a:b] += (amp * np.exp(2j * np.pi * rng.uniform(-0.5, 0.5) * np.arange(b - a))).astype(np.complex64)
    want = ofdm.ofdm_demod(opt).work(iq)
    s = ofdm.ofdm_demod(opt)
    got, pos, cuts = [], 0, []
    style = int(rng.integers(0, 3))
    while pos < len(iq):
        n = int(rng.integers(1, 600000)) if style == 0 else (int(rng.integers(1, 5000)) if (style == 1 and rng.random() < 0.5) else int(rng.integers(100000, 900000)))
        cuts.append(n)
        got += s.feed(iq[pos:pos + n])
        pos += n
    got += s.flush()
    ncase += 1
    if got != want:
        nbad += 1
        if nbad <= 3:
            out_dir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out_dir, exist_ok=True)
            np.savez_compressed(os.path.join(out_dir, "stream_fail_%d.npz" % nbad), iq=iq, cuts=np.array(cuts), mod=mod, N=N, occ=occ, CP=CP)
        nd = sum(1 for a, b in zip(got, want) if a != b)
        print("MISMATCH", json.dumps(dict(mod=mod, N=N, npkt=npkt, len=len(iq), snr=snr, cfo=cfo, got=len(got), want=len(want), differing=nd,
                                          okgot=sum(o for o, _ in got), okwant=sum(o for o, _ in want), cuts=cuts[:12])), flush=True)
print("fuzz_stream: %d cases, %d mismatches, seed %d" % (ncase, nbad, seed))
