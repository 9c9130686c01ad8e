#!/usr/bin/env python3
"""Stage-by-stage parity of the HIP engine against the CPU oracle (run on the GPU box).
Prints one line per stage; used while bringing kernels up.  tests/ holds the asserting version."""
import os, sys, struct, time, traceback
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
from ofdm_uhd_amd import config, options, _abi, engine
from oracle import oracle as orc


def run(mod, N, occ, CP, plen, npkt, snr_db=30.0, cfo_bins=0.0, seed=1):
    print("=== %s N=%d occ=%d CP=%d payload=%d npkt=%d snr=%g cfo=%g" % (mod, N, occ, CP, plen, npkt, snr_db, cfo_bins), flush=True)
    opt = options.default_options(modulation=mod, fft_length=N, occupied_tones=occ, cp_length=CP)
    cfg = config.make_cfg(opt)
    eng = engine.Engine(cfg=cfg)
    rng = np.random.default_rng(seed)
    pay = [struct.pack('!HH', i & 0xffff, 0) + rng.integers(0, 256, plen - 4, dtype=np.uint8).tobytes() for i in range(npkt)]
    # --- framing
    fr_g = eng.make_packets(pay)
    fr_o = [orc.make_packet(cfg, p) for p in pay]
    print("framing equal:", fr_g == fr_o, flush=True)
    # --- TX
    eng.set_taps(_abi.TAP_TX_FREQ)
    eng.set_channel(enable=False)
    iq_g = eng.tx(pay)
    iq_o, freq_o, _ = orc.tx(cfg, pay, want_taps=True)
    freq_g = eng.tap(_abi.TAP_TX_FREQ)
    print("tx freq maxerr:", float(np.abs(freq_g - freq_o).max()), " iq len", len(iq_g), len(iq_o),
          " iq maxerr:", float(np.abs(iq_g - iq_o).max()) if len(iq_g) == len(iq_o) else "LEN", flush=True)
    # --- channel (oracle output is THE input of both receivers)
    lead, tail = 2 * N, (N + CP) + 2 * N
    x = np.zeros(lead + len(iq_o) + tail, np.complex64)
    x[lead:lead + len(iq_o)] = iq_o
    psig = float(np.mean(np.abs(iq_o) ** 2))
    sigma = float(np.sqrt(psig / (10 ** (snr_db / 10.0))))
    cfo = cfo_bins * 2 * np.pi / N
    xg = eng.channel(x, sigma=sigma, cfo=cfo)
    orc.channel(x, sigma=sigma, cfo=cfo)
    print("channel maxerr:", float(np.abs(xg - x).max()), flush=True)
    # --- RX
    taps = (_abi.TAP_RX_CHAN_FILT, _abi.TAP_RX_METRIC, _abi.TAP_RX_FFT, _abi.TAP_RX_ACQ, _abi.TAP_RX_SINK, _abi.TAP_RX_PACKETS)
    mask = 0
    for t in taps:
        mask |= 1 << t
    t0 = time.time(); ro = orc.rx(cfg, x, mask); t1 = time.time()
    eng.set_taps(*taps)
    try:
        pk_g = eng.rx(x)
    except Exception as e:
        print("ENGINE RX FAILED:", e, eng.last_stats, flush=True)
        pk_g = None
    t2 = time.time()
    print("oracle rx %.3fs  engine rx %.3fs" % (t1 - t0, t2 - t1))
    print("oracle stats", ro.stats)
    print("engine stats", eng.last_stats, flush=True)
    y_o = ro.tap(_abi.TAP_RX_CHAN_FILT); y_g = eng.tap(_abi.TAP_RX_CHAN_FILT)
    print("chan_filt bit-exact:", bool(np.array_equal(y_o, y_g)), " maxerr", float(np.abs(y_o - y_g).max()), flush=True)
    u_o = ro.tap(_abi.TAP_RX_METRIC); u_g = eng.tap(_abi.TAP_RX_METRIC)
    ne = int(np.sum(u_o != u_g))
    print("metric bit-exact:", ne == 0, " mismatches", ne, " maxerr", float(np.abs(u_o - u_g).max()),
          " first", (int(np.flatnonzero(u_o != u_g)[0]) if ne else -1), flush=True)
    p_o = ro.tap(_abi.TAP_RX_PEAKS); p_g = eng.tap(_abi.TAP_RX_PEAKS)
    print("peaks equal:", list(p_o) == list(p_g), len(p_o), len(p_g), list(p_o[:6]), list(p_g[:6]), flush=True)
    a_o = ro.tap(_abi.TAP_RX_ANGLES); a_g = eng.tap(_abi.TAP_RX_ANGLES)
    if len(a_o) == len(a_g) and len(a_o):
        print("angles maxerr:", float(np.abs(a_o - a_g).max()))
    f_o = ro.tap(_abi.TAP_RX_FRAMES); f_g = eng.tap(_abi.TAP_RX_FRAMES)
    print("frames equal:", f_o.tolist() == f_g.tolist(), f_o[:4].tolist(), f_g[:4].tolist(), flush=True)
    for name, tp in (("fft", _abi.TAP_RX_FFT), ("acq", _abi.TAP_RX_ACQ), ("sink", _abi.TAP_RX_SINK)):
        a = ro.tap(tp); b = eng.tap(tp)
        if a.shape == b.shape and a.size:
            err = np.abs(a - b)
            print("%s shape %s maxerr %.3g  (max |ref| %.3g) worst row %d" % (name, a.shape, float(err.max()), float(np.abs(a).max()), int(err.max(axis=1).argmax())), flush=True)
        else:
            print("%s SHAPE MISMATCH" % name, a.shape, b.shape, flush=True)
    r_o = ro.tap(_abi.TAP_RX_PACKETS); r_g = eng.tap(_abi.TAP_RX_PACKETS)
    print("raw messages equal:", r_o.tobytes() == r_g.tobytes(), len(r_o), len(r_g))
    if pk_g is not None:
        print("packets equal:", pk_g == ro.packets, " n", len(pk_g), len(ro.packets), " ok", sum(ok for ok, _ in pk_g), sum(ok for ok, _ in ro.packets))
        print("payload recovery vs sent:", sum(1 for (ok, p) in pk_g if ok and p in pay), "/", npkt, flush=True)
    eng.close()


if __name__ == "__main__":
    cases = [
        ("qpsk", 512, 200, 128, 1026, 6, 30.0, 0.0),
        ("bpsk", 512, 200, 128, 300, 5, 30.0, 0.05),
        ("qpsk", 512, 200, 128, 1026, 6, 30.0, 0.3),
        ("8psk", 256, 120, 64, 500, 4, 30.0, 0.0),
        ("qam16", 2048, 1200, 512, 4091, 3, 30.0, 0.0),
        ("qam64", 1024, 600, 256, 2000, 3, 30.0, 0.1),
        ("qam64", 4096, 2400, 1024, 4091, 3, 30.0, 0.0),
        ("qam256", 64, 48, 16, 100, 4, 35.0, 0.0),
        ("bpsk", 128, 64, 32, 64, 4, 30.0, 0.0),
    ]
    sel = sys.argv[1:]
    for i, c in enumerate(cases):
        if sel and str(i) not in sel:
            continue
        try:
            run(*c)
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()
