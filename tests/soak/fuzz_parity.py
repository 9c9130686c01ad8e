"""Randomised parity soak: engine vs oracle over random geometries, constellations, packet mixes, SNR,
carrier offsets, carrier maps and stream lengths.  python tests/soak/fuzz_parity.py [seconds] [seed]"""
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
maps = [b["carrier_map"] for b in json.load(open(os.path.join(ROOT, "tests/golden/sense_blocks.json")))["blocks"]]
t_end = time.time() + budget
ncase = nbad = nlit = nmiss = 0
while time.time() < t_end:
    N = int(rng.choice([64, 128, 256, 512, 512, 512, 1024, 2048, 4096]))
    # multiples of 4 and, now and then, of 2: partial-nibble carrier maps (mapper and sink index rules differ)
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
    # detector / sampler parameters away from their defaults now and then
    cfg.sampler_timeout = int(rng.choice([1000, 1000, 1000, 7, 2]))
    if rng.random() < 0.25:
        cfg.peak_rise, cfg.peak_fall = float(rng.choice([0.1, 0.3, 0.5])), float(rng.choice([0.1, 0.2, 0.4]))
    if rng.random() < 0.2:
        cfg.peak_alpha = float(rng.choice([0.01, 0.0003]))
    if rng.random() < 0.2:
        cfg.max_fft_shift_len = int(rng.choice([1, 2, 8]))
    npkt = int(rng.integers(1, 12))
    sizes = rng.integers(0, min(4091, 40 * N), npkt)
    pay = make_payloads(npkt, sizes, seed=int(rng.integers(0, 1 << 30)))
    lead = int(rng.integers(0, 3 * N + 50))
    tail = int(rng.integers(0, 4 * N + 2 * CP))
    snr = float(rng.choice([12.0, 20.0, 30.0, 40.0, 60.0, 80.0, 100.0]))
    cfo = float(rng.choice([0.0, 0.0, 0.03, -0.2, 0.45, 1.3, -3.2]))
    desc = dict(N=N, occ=occ, CP=CP, mod=mod, carriers=carriers, npkt=npkt, sizes=sizes.tolist(), lead=lead, tail=tail, snr=snr, cfo=cfo,
                pad=pad_usrp, timeout=int(cfg.sampler_timeout), rise=float(cfg.peak_rise), fall=float(cfg.peak_fall),
                alpha=float(cfg.peak_alpha), shift=int(cfg.max_fft_shift_len))
    try:
        eng = engine.Engine(cfg=cfg)
    except ValueError:
        continue
    ncase += 1
    try:
        # TX
        try:
            iq_o, freq_o, framed_o = orc.tx(cfg, pay, want_taps=True)
        except ValueError:
            # an illegal batch (USRP padding pushes a maximum-length packet past the whitening mask): the
            # engine must refuse it the same way
            try:
                eng.tx(pay)
                raise AssertionError("engine accepted a batch the oracle refuses")
            except ValueError:
                continue
        assert eng.make_packets(pay) == framed_o, "framed packets"
        eng.set_taps(_abi.TAP_TX_FREQ)
        iq_g = eng.tx(pay)
        assert np.array_equal(eng.tap(_abi.TAP_TX_FREQ), freq_o), "tx freq"
        assert np.array_equal(iq_g, iq_o), "tx iq"
        # channel + RX
        x = np.concatenate([np.zeros(lead, np.complex64), iq_o, np.zeros(tail, np.complex64)])
        if rng.random() < 0.3 and len(iq_o) > 4 * (N + CP):
            # a silent gap cut into the burst at a symbol boundary (energy drops by the full SNR inside a tile)
            L_ = N + CP
            cut = lead + int(rng.integers(1, len(iq_o) // L_)) * L_
            x = np.concatenate([x[:cut], np.zeros(int(rng.integers(1, 6000)), np.complex64), x[cut:]])
            desc["gap_at"] = cut
        core = iq_o if len(iq_o) else np.ones(1, np.complex64)
        sigma = float(np.sqrt(np.mean(np.abs(core) ** 2) / 10 ** (snr / 10)))
        orc.channel(x, sigma=sigma, cfo=cfo * 2 * np.pi / N, seed=int(rng.integers(0, 1 << 30)))
        if rng.random() < 0.15 and len(x):
            # a carrier on top (anything from buried in the noise to 20 dB above the signal), sometimes only on
            # a stretch: drives the metric above the candidate threshold over long runs
            amp = float(np.sqrt(np.mean(np.abs(core) ** 2)) * 10 ** rng.uniform(-2, 1))
            a, b = (0, len(x)) if rng.random() < 0.5 else sorted(rng.integers(0, len(x), 2).tolist())
            x[a:b] += (amp * np.exp(2j * np.pi * rng.uniform(-0.5, 0.5) * np.arange(b - a))).astype(np.complex64)
            desc["carrier"] = [amp, int(a), int(b)]
        taps = (_abi.TAP_RX_CHAN_FILT, _abi.TAP_RX_METRIC, _abi.TAP_RX_PRESEL, _abi.TAP_RX_SAMPLER, _abi.TAP_RX_FFT, _abi.TAP_RX_ACQ, _abi.TAP_RX_SINK,
                _abi.TAP_RX_PACKETS)
        mask = 0
        for t in taps:
            mask |= 1 << t
        ro = orc.rx(cfg, x, mask)
        eng.set_taps(*taps)
        pk = eng.rx(x)
        pg, po = eng.tap(_abi.TAP_RX_PEAKS).tolist(), ro.tap(_abi.TAP_RX_PEAKS).tolist()
        if pg != po:
            u = ro.tap(_abi.TAP_RX_METRIC)
            d = sorted(set(pg) ^ set(po))[:6]
            print("peaks differ: gpu %d oracle %d; symmetric difference (first) %s" % (len(pg), len(po), d))
            for q0 in d[:3]:
                w = u[max(0, q0 - 6):q0 + 7]
                print("   u around %d (in gpu: %s, in oracle: %s): %s" % (q0, q0 in pg, q0 in po, np.array2string(w, precision=7)))
        assert pg == po, "peaks"
        if po != ro.tap(orc.TAP_PEAKS_GR).tolist():
            nlit += 1          # (the literal float32 recurrence differs: a marginal run, both sides the same)
        nmiss += 1 if ro.presel_miss else 0
        assert eng.tap(_abi.TAP_RX_FRAMES).tolist() == ro.tap(_abi.TAP_RX_FRAMES).tolist(), "frames"
        yg, yo = eng.tap(_abi.TAP_RX_CHAN_FILT), ro.tap(_abi.TAP_RX_CHAN_FILT)
        if not np.array_equal(yg, yo):
            bad = np.flatnonzero(yg != yo)
            print("chan_filt: %d of %d samples differ, first %s; gpu %s oracle %s; input there %s" % (
                len(bad), len(yo), bad[:8].tolist(), yg[bad[:3]], yo[bad[:3]], x[bad[:3]]))
        assert np.array_equal(yg, yo), "chan_filt"
        assert np.array_equal(eng.tap(_abi.TAP_RX_METRIC), ro.tap(_abi.TAP_RX_METRIC)), "metric"
        assert np.array_equal(eng.tap(_abi.TAP_RX_PRESEL), ro.tap(_abi.TAP_RX_PRESEL), equal_nan=True), "presel"
        assert np.array_equal(eng.tap(_abi.TAP_RX_ANGLES), ro.tap(_abi.TAP_RX_ANGLES)), "angles"
        for k in ("symbols", "peaks", "frames"):
            assert eng.last_stats[k] == ro.stats[k], k
        # every float stage carries the oracle's bits: no tolerance, no exemption
        for tap, name in ((_abi.TAP_RX_SAMPLER, "sampler"), (_abi.TAP_RX_FFT, "fft"), (_abi.TAP_RX_ACQ, "acq"),
                          (_abi.TAP_RX_SINK, "sink")):
            a, b = ro.tap(tap), eng.tap(tap)
            assert a.shape == b.shape, name + " shape"
            # NaN where a coarse offset beyond the guard band makes the equaliser divide by an empty bin: same places
            assert np.array_equal(a, b, equal_nan=True), name
        assert pk == ro.packets, "packets"
        assert eng.tap(_abi.TAP_RX_PACKETS).tobytes() == ro.tap(_abi.TAP_RX_PACKETS).tobytes(), "raw messages"
        for k in ("headers_ok", "packets", "crc_ok", "chained_frames"):
            assert eng.last_stats[k] == ro.stats[k], k
    except (AssertionError, engine.EngineError, ValueError) as e:
        nbad += 1
        if os.environ.get("FUZZ_STOP"):
            t_end = 0
        print("MISMATCH [%s] case %d" % (e if isinstance(e, AssertionError) else "error " + str(e)[:80], ncase), json.dumps(desc), flush=True)
        # keep what is needed to replay the case without the random sequence that led to it
        try:
            out = os.path.join(ROOT, "gpurun_out", "fuzz_parity_fail_%d_%d.npz" % (seed, ncase))
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savez_compressed(out, x=x, desc=json.dumps(desc), payloads=np.array([p.hex() for p in pay]))
            print("   capture saved to", out, flush=True)
        except Exception as e2:
            print("   (capture not saved: %s)" % e2, flush=True)
    finally:
        eng.close()
print("fuzz: %d cases, %d mismatches, seed %d (captures on which the literal recurrence differs: %d; with a pre-selection miss: %d)" % (
    ncase, nbad, seed, nlit, nmiss))
