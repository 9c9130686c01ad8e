"""Replay a capture saved by tests/soak/fuzz_stream.py and show where feed() and the one-shot call part ways."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ofdm_uhd_amd import ofdm, options, _abi
d = np.load(sys.argv[1])
iq, cuts = d["iq"], d["cuts"].tolist()
mod, N, occ, CP = str(d["mod"]), int(d["N"]), int(d["occ"]), int(d["CP"])
opt = options.default_options(modulation=mod, fft_length=N, occupied_tones=occ, cp_length=CP)
one = ofdm.ofdm_demod(opt)
want = one.work(iq)
wpos = one.engine().rx_packet_pos().astype(np.int64)
fl1, phi1, st1, _sw = one.engine().rx_nco_state()
print("one-shot: %d packets, %d flags" % (len(want), len(fl1)))
s = ofdm.ofdm_demod(opt)
T, span, lookback = s._stream_geometry()
print("T", T, "span", span, "lookback", lookback, "len", len(iq), "cuts", cuts)
got, gpos, pos = [], [], 0
for n in cuts + [0]:
    flush = n == 0
    base_before = getattr(s, "_s_abs", 0)
    out = s.feed(iq[pos:pos + n], flush=flush)
    pos += n
    print("call: fed %d flush %s -> %d packets; base %d final %s hist %d" % (n, flush, len(out), base_before, getattr(s, "_s_final", None), len(getattr(s, "_s_hist", []))))
    got += out
i = 0
while i < min(len(got), len(want)) and got[i] == want[i]:
    i += 1
print("first difference at packet", i, "of", len(got), len(want))
for k in range(max(0, i - 2), min(len(want), i + 3)):
    print(" want", k, want[k][0], len(want[k][1]), "pos", int(wpos[k]))
for k in range(max(0, i - 2), min(len(got), i + 3)):
    print(" got ", k, got[k][0], len(got[k][1]))
print("one-shot flags near:", [int(f) for f in fl1 if abs(int(f) - int(wpos[min(i, len(wpos) - 1)])) < 300000])

# ---- detail: replay, and after every call compare its flags / phases / steps / K with the one-shot's
print("\n--- per-call NCO state vs one-shot")
eng1 = one.engine()
eng1.set_taps(_abi.TAP_RX_FRAMES)
eng1.set_flag_history(None)
eng1.rx(iq)
fl1, ph1, st1, _sw1 = eng1.rx_nco_state()
fr1 = {int(a): int(b) for a, b in eng1.tap(_abi.TAP_RX_FRAMES)}
ref = {int(f): (int(p), float(s)) for f, p, s in zip(fl1, ph1, st1)}
s = ofdm.ofdm_demod(opt)
s.engine().set_taps(_abi.TAP_RX_FRAMES)
pos = 0
for n in cuts + [0]:
    base = getattr(s, "_s_abs", 0)
    s.feed(iq[pos:pos + n], flush=(n == 0))
    pos += n
    fl, ph, st, _sw2 = s.engine().rx_nco_state()
    fr = {int(a) + base: int(b) for a, b in s.engine().tap(_abi.TAP_RX_FRAMES)} if len(fl) else {}
    bad = []
    for f, p, q in zip(fl, ph, st):
        a = int(f) + base
        if a not in ref:
            bad.append((a, "not a one-shot flag"))
        elif ref[a] != (int(p), float(q)):
            bad.append((a, "phase/step differ", ref[a], (int(p), float(q))))
        if a in fr1 and a in fr and fr1[a] != fr[a]:
            bad.append((a, "K differs", fr1[a], fr[a]))
    print("call base", base, "flags", len(fl), "first", int(fl[0]) + base if len(fl) else None, "issues:", bad[:6])

print("\n--- raw rx() output per call around the difference")
s = ofdm.ofdm_demod(opt)
pos = 0
orig_rx = s.engine().rx
def spy(buf, *a, **k):
    r = orig_rx(buf, *a, **k)
    base = s._s_abs
    pp = s.engine().rx_packet_pos().astype(np.int64) + base
    print(" rx(base %d, len %d): " % (base, len(buf)) + ", ".join("%d:%s/%d" % (int(p), "ok" if ok else "bad", len(pl)) for (ok, pl), p in zip(r, pp) if 450000 < p < 700000),
          "| stats", {k2: s.engine().last_stats[k2] for k2 in ("frames", "headers_ok", "packets", "chained_frames")})
    return r
s.engine().rx = spy
for n in cuts + [0]:
    s.feed(iq[pos:pos + n], flush=(n == 0))
    pos += n
r = one.engine().rx(iq)
pp = one.engine().rx_packet_pos().astype(np.int64)
print(" one-shot: " + ", ".join("%d:%s/%d" % (int(p), "ok" if ok else "bad", len(pl)) for (ok, pl), p in zip(r, pp) if 450000 < p < 700000),
      "| stats", {k2: one.engine().last_stats[k2] for k2 in ("frames", "headers_ok", "packets", "chained_frames")})
