"""Spectrum sensor on the GPU (csrc/sense.h) through the C ABI, against the oracle and the
reference's recorded sensing blocks."""
import json
import os

import numpy as np
import pytest

from helpers import loopback_stream, make_cfg, make_payloads
from ofdm_uhd_amd import config, engine, predictive_sense

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(cfg=make_cfg("qpsk"))
    yield e
    e.close()


def _stream(S, nvec, seed, tones=((0.25, 0.03),), floor=1e-6):
    """Noise with a per-bin power floor of `floor` plus tones given as (normalised frequency,
    sqrt of the peak bin power): levels are set relative to the 1e-4 threshold for every S."""
    rng = np.random.default_rng(seed)
    n = nvec * S + 7
    sigma = np.sqrt(floor / (0.52 * S))         # sum(w^2) ~ 0.26 S for Blackman-Harris-92
    iq = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * sigma).astype(np.complex64)
    t = np.arange(n)
    for f, a in tones:
        iq += (a / (0.36 * S) * np.exp(2j * np.pi * (round(f * S) + 0.3) * t / S)).astype(np.complex64)
    return iq


def _compare(got, ref, sc):
    S = sc.fft_size
    assert got["msgs"].shape == ref["msgs"].shape
    # bar: 1e-5 of the spectrum's peak (north star).  Engine and oracle run the same window product, the same
    # transform schedule and the same |.|^2, and a max is order-independent: the message bodies are bit-identical,
    # and with them the float64 means, the decisions and the hex maps
    assert np.array_equal(got["msgs"], ref["msgs"])
    assert got["mean"].shape == ref["mean"].shape
    assert np.array_equal(got["mean"], ref["mean"])
    assert np.array_equal(got["bits"], ref["bits"])
    assert got["hex"] == ref["hex"]
    for d, h in enumerate(got["hex"]):
        assert h == predictive_sense.hex_conv(got["bits"][d].tolist())
    # the tail itself is exact arithmetic on the GPU's own messages: float64 sum in message
    # order / avg, threshold, half swap
    per = sc.avg_msgs + sc.skip_msgs
    for d in range(len(got["hex"])):
        acc = np.zeros(S, np.float64)
        for k in range(sc.avg_msgs):
            acc = acc + got["msgs"][d * per + k].astype(np.float64)
        acc = acc / float(sc.avg_msgs)
        ino = np.concatenate([acc[S // 2:], acc[:S // 2]])
        assert np.array_equal(got["mean"][d], ino)
        assert np.array_equal(got["bits"][d], (ino <= sc.threshold).astype(np.uint8))


@pytest.mark.parametrize("S,tune,dwell,avg", [
    (64, 0, 3, 2), (128, 1, 40, 3), (256, 24, 244, 10), (512, 2, 9, 3), (1024, 1, 17, 2), (2048, 0, 5, 2), (4096, 1, 3, 2)])
def test_sense_matches_oracle(eng, orc, S, tune, dwell, avg):
    sc = config.make_sense_cfg(S, tune, dwell, avg, 1, threshold=1e-4)
    nmsg = 2 * (avg + 1) + 1
    iq = _stream(S, (tune + dwell) * nmsg + 1, seed=S, tones=((0.25, 0.03), (0.7, 0.017)))
    got = eng.sense(sc, iq)
    ref = orc.sense(sc, iq)
    assert len(got["hex"]) == 2 and len(got["msgs"]) == nmsg
    _compare(got, ref, sc)
    # the strong tone is flagged occupied, the quiet quarter is free
    k = (round(0.25 * S) + S // 2) % S
    assert got["bits"][0][k] == 0 and got["bits"][0][(k + S // 2) % S] == 1


def test_sense_edge_cases(eng, orc):
    sc = config.make_sense_cfg(256, 2, 3, 10, 1)
    # too short for one message / exactly one message / empty
    for n in (0, 5 * 256 - 1, 5 * 256):
        iq = _stream(256, 6, 3)[:n]
        got = eng.sense(sc, iq)
        assert got["msgs"].shape[0] == n // (5 * 256) and got["hex"] == []
        if n // (5 * 256):
            assert np.array_equal(got["msgs"], orc.sense(sc, iq)["msgs"])
    # max-hold really is a max over the dwell: one loud vector among quiet ones decides the message
    iq = _stream(256, 5 * 11, 4, tones=())
    loud = (0.001 * np.exp(2j * np.pi * 40 * np.arange(256) / 256)).astype(np.complex64)
    iq[3 * 256:4 * 256] += loud            # vector 3 = 2nd accrued vector of message 0
    iq[5 * 256:6 * 256] += loud * 3        # vector 5 = inside message 1's tune delay: ignored
    got = eng.sense(sc, iq)
    assert got["msgs"][0][40] > 5e-3 and got["msgs"][1][40] < 1e-4
    _compare(got, orc.sense(sc, iq), sc)
    # NaN input never poisons the max (accrue_stats' `>` test ignores it), Inf does propagate
    iq2 = iq.copy()
    iq2[9 * 256 + 5] = np.nan
    got2 = eng.sense(sc, iq2)
    ref2 = orc.sense(sc, iq2)
    assert np.isfinite(got2["msgs"]).all() and np.isfinite(ref2["msgs"]).all()
    with pytest.raises(ValueError):
        bad = config.make_sense_cfg(256)
        bad.dwell_delay = 0
        eng.sense(bad, iq)


def test_sense_decide_recorded_blocks(eng, orc):
    """GPU sense_loop tail vs the 43 blocks the reference recorded (pinned parity)."""
    with open(os.path.join(HERE, "golden", "sense_blocks.json")) as f:
        blocks = json.load(f)["blocks"]
    sc = config.make_sense_cfg()
    S = 256
    msgs = []
    for b in blocks:
        p = np.array([float(v) for v in b["power"]], np.float64)
        m = np.tile(np.concatenate([p[S // 2:], p[:S // 2]]).astype(np.float32), (11, 1))
        m[10] = 1.0e3
        msgs.append(m)
    msgs = np.concatenate(msgs)
    got = eng.sense_decide(sc, msgs)
    ref = orc.sense_decide(sc, msgs)
    assert got["hex"] == [b["carrier_map"] for b in blocks] == ref["hex"]
    assert np.array_equal(got["bits"], np.array([b["bit"] for b in blocks], np.uint8))
    assert np.array_equal(got["mean"], ref["mean"])  # float64 sums in the same order: bit-exact


def test_rx_with_fused_sensing(eng, orc):
    """BASELINE config 5 shape: the receiver and the sensor share one IQ buffer in one call."""
    cfg = eng.cfg
    payloads = make_payloads(24, 300, seed=9)
    iq = loopback_stream(orc, cfg, payloads, snr_db=30.0)
    # 30 dB SNR puts the out-of-band noise floor near 5e-3 per bin: threshold between it and the signal
    sc = config.make_sense_cfg(256, 2, 12, 3, 1, threshold=0.05)
    plain = eng.rx(iq)
    alone = eng.sense(sc, iq)
    eng.set_rx_sense(sc)
    try:
        fused_pk = eng.rx(iq)
        fused = eng.rx_sense_result(len(iq))
        # second call: buffers are reused
        fused_pk2 = eng.rx(iq)
        fused2 = eng.rx_sense_result(len(iq))
    finally:
        eng.set_rx_sense(None)
    assert fused_pk == plain == fused_pk2 and len(plain) == 24 and all(ok for ok, _ in plain)
    for k in ("msgs", "mean", "bits"):
        assert np.array_equal(fused[k], alone[k]) and np.array_equal(fused2[k], alone[k])
    assert fused["hex"] == alone["hex"] == fused2["hex"] and len(alone["hex"]) >= 1
    _compare(fused, orc.sense(sc, iq), sc)
    # the OFDM signal occupies the middle 200/512 of the band: those bins are flagged busy
    bits = fused["bits"][0]
    assert bits[128 - 40:128 + 40].sum() == 0 and bits[:20].sum() == 20 and bits[-20:].sum() == 20
    assert eng.rx(iq) == plain  # switched off again


def test_rx_with_fused_sensing_c5_size(orc):
    """BASELINE config 5 sizing proper: N=4096 / 64-QAM / CP=1024 receiver with a 4096-point sensor riding on the
    same IQ buffer in the same call."""
    cfg = make_cfg("qam64", 4096, 2400, 1024)
    e = engine.Engine(cfg=cfg)
    payloads = make_payloads(6, 4091, seed=12)
    iq = loopback_stream(orc, cfg, payloads, snr_db=40.0)
    sc = config.make_sense_cfg(4096, 0, 2, 2, 1, threshold=0.05)
    plain = e.rx(iq)
    assert plain == orc.rx(cfg, iq).packets and sum(ok for ok, _ in plain) >= 4
    alone = e.sense(sc, iq)
    e.set_rx_sense(sc)
    try:
        fused_pk = e.rx(iq)
        fused = e.rx_sense_result(len(iq))
    finally:
        e.set_rx_sense(None)
    assert fused_pk == plain
    for k in ("msgs", "mean", "bits"):
        assert np.array_equal(fused[k], alone[k])
    assert fused["hex"] == alone["hex"] and len(alone["hex"]) >= 1
    _compare(fused, orc.sense(sc, iq), sc)
    # the signal sits in the middle 2400/4096 of the band
    bits = fused["bits"][0]
    assert bits[2048 - 600:2048 + 600].sum() == 0 and bits[:300].sum() == 300
    e.close()


def test_sensor_mirror(eng, tmp_path):
    """predictive_sense.sensor / sense_loop driven the reference's way (-i FILE)."""
    from ofdm_uhd_amd import iqio
    S = 256
    iq = _stream(S, (24 + 244) * 11 + 1, seed=11, tones=((0.1, 0.03),))
    f = str(tmp_path / "sense.dat")
    iqio.file_sink(f).write(iq)
    tb = predictive_sense.sensor(["-i", f, "-p", "895M", "-q", "905M"], engine=eng)
    fifo = open(str(tmp_path / "fifo"), "wb")
    out = predictive_sense.sense_loop(tb, fifo=fifo)
    fifo.close()
    assert len(out) == 1
    e = out[0]
    assert len(e["hexa_thr"]) == 64 and open(str(tmp_path / "fifo"), "rb").read().decode() == e["hexa_thr"]
    assert e["ofdm_center_freq"] == 900e6 and e["sensed_freq"][127] == 900e6
    assert predictive_sense.hex_conv(e["thrshold_inorder"]) == e["hexa_thr"]
    k = (round(0.1 * S) + S // 2) % S
    assert e["thrshold_inorder"][k] == 0 and sum(e["thrshold_inorder"]) > 200
    msgs = tb.messages()
    assert len(msgs) == 11 and msgs[0].vlen == 256 and len(msgs[0].raw_data) == 1024
