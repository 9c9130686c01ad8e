"""Continuous operation: ofdm_demod.feed() over arbitrary chunks == one call on the whole capture."""
import numpy as np
import pytest

from helpers import make_cfg, make_payloads
from ofdm_uhd_amd import ofdm, options

pytestmark = pytest.mark.gpu


def _capture(orc, mod="qpsk", npkt=90, seed=5, cfo_bins=0.07, snr_db=30.0):
    """Bursts of packets of mixed sizes separated by silences, one maximum-length packet, AWGN + CFO."""
    cfg = make_cfg(mod)
    rng = np.random.default_rng(seed)
    sizes = rng.integers(20, 1500, npkt)
    sizes[npkt // 3] = 4091                                     # largest legal packet
    pay = make_payloads(npkt, sizes, seed=seed)
    parts, k = [np.zeros(1500, np.complex64)], 0
    while k < npkt:
        n = int(rng.integers(1, 9))
        parts.append(orc.tx(cfg, pay[k:k + n]))
        parts.append(np.zeros(int(rng.integers(200, 60000)), np.complex64))
        k += n
    iq = np.concatenate(parts)
    psig = float(np.mean(np.abs(parts[1]) ** 2))
    orc.channel(iq, sigma=float(np.sqrt(psig / 10 ** (snr_db / 10))), cfo=cfo_bins * 2 * np.pi / cfg.fft_length, seed=77)
    return cfg, pay, iq


def _demod(mod="qpsk"):
    return ofdm.ofdm_demod(options.default_options(modulation=mod))


@pytest.mark.parametrize("chunking", ["131072", "50000", "random", "tiny-then-big", "one"])
def test_feed_equals_one_shot(orc, chunking):
    cfg, pay, iq = _capture(orc)
    d = _demod()
    want = d.work(iq)
    good = [p for ok, p in want if ok]
    # the capture is sound (the reference's receiver loses the first packet of a burst: its preamble is
    # derotated with the frequency estimate of the noise before it)
    assert len(good) >= len(pay) * 0.6 and all(p in pay for p in good)
    rng = np.random.default_rng(1)
    cuts, pos = [], 0
    while pos < len(iq):
        if chunking == "random":
            n = int(rng.integers(1, 400000))
        elif chunking == "tiny-then-big":
            n = 1000 if pos < 20000 else 700000
        elif chunking == "one":
            n = len(iq)
        else:
            n = int(chunking)
        cuts.append((pos, min(len(iq), pos + n)))
        pos += n
    got = []
    s = _demod()
    fired = []
    s._callback = lambda ok, p: fired.append((ok, p))
    for a, b in cuts:
        got += s.feed(iq[a:b])
    got += s.flush()
    assert got == want
    assert fired == want                                       # the callback sees the same sequence
    # the stream state is gone after flush(): a second capture starts afresh
    assert s.feed(iq[:200000], flush=True) == d.work(iq[:200000])


def test_feed_other_geometry(orc):
    """Larger FFT / other constellation: the carry lengths follow the configuration."""
    cfg = make_cfg("qam16", 2048, 1200, 512)
    pay = make_payloads(12, 3000, seed=2)
    parts = [np.zeros(5000, np.complex64)]
    for i in range(0, 12, 3):
        parts += [orc.tx(cfg, pay[i:i + 3]), np.zeros(30000, np.complex64)]
    iq = np.concatenate(parts)
    orc.channel(iq, sigma=float(np.sqrt(np.mean(np.abs(parts[1]) ** 2) / 10 ** 4.0)), seed=3)
    opt = options.default_options(modulation="qam16", fft_length=2048, occupied_tones=1200, cp_length=512)
    want = ofdm.ofdm_demod(opt).work(iq)
    s = ofdm.ofdm_demod(opt)
    got = []
    for a in range(0, len(iq), 250000):
        got += s.feed(iq[a:a + 250000])
    got += s.flush()
    assert got == want and sum(ok for ok, _ in want) >= 8


def test_feed_after_reset_carrier_map(orc):
    """reset_carrier_map to a sparse sensed map (packets then take up to occ/ncar times more symbols) and then
    feed() in chunks: the carried span must follow the live map, or a long packet is delivered truncated."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sense_blocks.json")) as f:
        maps = [b["carrier_map"][:50] for b in json.load(f)["blocks"]]
    carriers = min(maps, key=lambda m: sum(bin(int(c, 16)).count("1") for c in m))   # the sparsest recorded map
    ncar = sum(bin(int(c, 16)).count("1") for c in carriers)
    assert 0 < ncar < 150
    cfg = make_cfg("qpsk", 512, 200, 128, carriers=carriers)
    pay = make_payloads(8, [4091, 300, 4091, 1500, 4091, 20, 2500, 4091], seed=11)
    parts = [np.zeros(3000, np.complex64)]
    for i in range(0, 8, 2):
        parts += [orc.tx(cfg, pay[i:i + 2]), np.zeros(20000, np.complex64)]
    iq = np.concatenate(parts)
    orc.channel(iq, sigma=float(np.sqrt(np.mean(np.abs(parts[1]) ** 2) / 10 ** 3.0)), seed=5)
    d = _demod()
    d.reset_carrier_map(carriers)
    want = d.work(iq)
    assert want == orc.rx(cfg, iq).packets and sum(ok for ok, _ in want) >= 5
    s = _demod()
    s.reset_carrier_map(carriers)
    got = []
    for a in range(0, len(iq), 60000):          # shorter than one maximum-length packet on this map
        got += s.feed(iq[a:a + 60000])
    got += s.flush()
    assert got == want
