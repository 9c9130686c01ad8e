"""CPU-only replay of tests/soak/fuzz_parity.py's case sequence (same draws from the same seed) with the oracle alone:
prints every capture in which gr_peak_detector_fb run LITERALLY (float32 recurrence of the average from the first
sample) and its normative evaluation (what the engine runs) raise different flags -- the
"marginal" captures.  Each is printed as an explicit descriptor (all seeds included) that
tests/test_gpu_parity.py::test_marginal_detector_cases regenerates.
    python tests/soak/find_marginal.py <seed> <ncases> [want_N want_occ want_CP]"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_cfg, make_payloads
from ofdm_uhd_amd import _abi
from oracle import oracle as orc

seed = int(sys.argv[1]); ncases = int(sys.argv[2])
want = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else None
rng = np.random.default_rng(seed)
maps = [b["carrier_map"] for b in json.load(open(os.path.join(ROOT, "tests/golden/sense_blocks.json")))["blocks"]]


def engine_accepts(cfg):
    """what ofdm_create checks beyond config.make_cfg: the frame sink's carrier map"""
    m = np.zeros(_abi.OFDM_MAX_FFT, np.int32)
    car = cfg.carrier_map or b"FE7F"
    return orc.lib().orc_carrier_map2(cfg.occupied_tones, cfg.occupied_tones, car, 1, m.ctypes.data_as(C.c_void_p), len(m)) > 0


orc.lib().orc_carrier_map2.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
ncase = 0
while ncase < ncases:
    N = int(rng.choice([64, 128, 256, 512, 512, 512, 1024, 2048, 4096]))
    occ = int(rng.integers(4, N // 4 - 1)) * 4 if N > 64 else int(rng.choice([16, 32, 48, 52, 60]))
    if rng.random() < 0.15:
        occ += 2
    occ = max(16, min(occ, N - 4))
    CP = int(rng.integers(1, max(2, N // 2)))
    if rng.random() < 0.5:
        CP = max(8, (CP // 8) * 8)
    mod = str(rng.choice(["bpsk", "qpsk", "qpsk", "8psk", "qam16", "qam64", "qam256"]))
    carriers = None
    if N == 512 and occ == 200 and rng.random() < 0.5:
        carriers = maps[int(rng.integers(0, len(maps)))][:50]
    pad_usrp = bool(rng.random() < 0.2)
    try:
        cfg = make_cfg(mod, N, occ, CP, carriers=carriers, pad_for_usrp=pad_usrp)
    except ValueError:
        continue
    cfg.sampler_timeout = int(rng.choice([1000, 1000, 1000, 7, 2]))
    if rng.random() < 0.25:
        cfg.peak_rise, cfg.peak_fall = float(rng.choice([0.1, 0.3, 0.5])), float(rng.choice([0.1, 0.2, 0.4]))
    if rng.random() < 0.2:
        cfg.peak_alpha = float(rng.choice([0.01, 0.0003]))
    if rng.random() < 0.2:
        cfg.max_fft_shift_len = int(rng.choice([1, 2, 8]))
    npkt = int(rng.integers(1, 12))
    sizes = rng.integers(0, min(4091, 40 * N), npkt)
    pseed = int(rng.integers(0, 1 << 30))
    pay = make_payloads(npkt, sizes, seed=pseed)
    lead = int(rng.integers(0, 3 * N + 50))
    tail = int(rng.integers(0, 4 * N + 2 * CP))
    snr = float(rng.choice([12.0, 20.0, 30.0, 40.0, 60.0, 80.0, 100.0]))
    cfo = float(rng.choice([0.0, 0.0, 0.03, -0.2, 0.45, 1.3, -3.2]))
    desc = dict(N=N, occ=occ, CP=CP, mod=mod, carriers=carriers, npkt=npkt, sizes=sizes.tolist(), pseed=pseed, lead=lead, tail=tail,
                snr=snr, cfo=cfo, pad=pad_usrp, timeout=int(cfg.sampler_timeout), rise=float(cfg.peak_rise),
                fall=float(cfg.peak_fall), alpha=float(cfg.peak_alpha), shift=int(cfg.max_fft_shift_len))
    if not engine_accepts(cfg):
        continue
    ncase += 1
    try:
        iq_o = orc.tx(cfg, pay)
    except ValueError:
        continue
    x = np.concatenate([np.zeros(lead, np.complex64), iq_o, np.zeros(tail, np.complex64)])
    if rng.random() < 0.3 and len(iq_o) > 4 * (N + CP):
        L_ = N + CP
        cut = lead + int(rng.integers(1, len(iq_o) // L_)) * L_
        glen = int(rng.integers(1, 6000))
        x = np.concatenate([x[:cut], np.zeros(glen, np.complex64), x[cut:]])
        desc["gap_at"], desc["gap_len"] = cut, glen
    core = iq_o if len(iq_o) else np.ones(1, np.complex64)
    sigma = float(np.sqrt(np.mean(np.abs(core) ** 2) / 10 ** (snr / 10)))
    cseed = int(rng.integers(0, 1 << 30))
    desc["sigma"], desc["cseed"] = sigma, cseed
    skip = want is not None and (N, occ, CP) != want
    if not skip:
        orc.channel(x, sigma=sigma, cfo=cfo * 2 * np.pi / N, seed=cseed)
    if rng.random() < 0.15 and len(x):
        amp = float(np.sqrt(np.mean(np.abs(core) ** 2)) * 10 ** rng.uniform(-2, 1))
        a, b = (0, len(x)) if rng.random() < 0.5 else sorted(rng.integers(0, len(x), 2).tolist())
        fr = float(rng.uniform(-0.5, 0.5))
        if not skip:
            x[a:b] += (amp * np.exp(2j * np.pi * fr * np.arange(b - a))).astype(np.complex64)
        desc["carrier"] = [amp, int(a), int(b), fr]
    if skip:
        continue
    ro = orc.rx(cfg, x, 0)
    p, pg = ro.tap(_abi.TAP_RX_PEAKS).tolist(), ro.tap(orc.TAP_PEAKS_GR).tolist()
    if want is not None:
        print("WANTED case %d: flags %d / literal %d  miss %d" % (ncase, len(p), len(pg), ro.presel_miss), json.dumps(desc), flush=True)
    if p != pg:
        print("MARGINAL case %d: flags %d / literal %d, differ at %s miss %d" % (ncase, len(p), len(pg), sorted(set(p) ^ set(pg))[:4], ro.presel_miss),
              json.dumps(desc), flush=True)
    elif ro.presel_miss:
        print("presel_miss %d in case %d (flags equal) N=%d CP=%d snr=%g" % (ro.presel_miss, ncase, N, CP, snr), flush=True)
    ro.close()
print("replayed %d cases of seed %d" % (ncase, seed))
