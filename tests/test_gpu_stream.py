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


def _read_logs(dirname):
    import os
    out = {}
    for name in sorted(os.listdir(dirname)):
        if name.endswith(".dat"):
            out[name] = np.fromfile(os.path.join(dirname, name), np.complex64)
    return out


RX_LOGS = ("ofdm_receiver-chan_filt_c.dat", "ofdm_receiver-fft_out_c.dat", "ofdm_receiver-frame_acq_c.dat",
           "ofdm_receiver-sampler_c.dat", "ofdm_receiver-sigmix_c.dat", "ofdm_receiver-nco_c.dat", "ofdm_frame_sink_c.dat")
TX_LOGS = ("ofdm_mapper_c.dat", "ofdm_preambles.dat", "ofdm_ifft_c.dat", "ofdm_cp_adder_c.dat")


def test_log_probe_files_accumulate(orc, tmp_path, monkeypatch):
    """--log (ofdm.py:123-131,253-254; ofdm_receiver.py~:144-152): the eleven probe files exist, hold the taps' bytes and
    ACCUMULATE like gr.file_sink does for the life of the flow graph -- across flushes of the modulator, across calls
    of the demodulator, and across the chunks of a fed stream, whose files equal those of one call on the capture."""
    import os
    from ofdm_uhd_amd import _abi
    monkeypatch.chdir(tmp_path)
    opt = options.default_options(modulation="qpsk", log=True)
    # --- modulator: two flushes append
    m = ofdm.ofdm_mod(opt, msgq_limit=4, pad_for_usrp=False)
    pay = make_payloads(6, 300, seed=12)
    for p in pay[:3]:
        m.send_pkt(p)
    a = m.flush()
    for p in pay[3:]:
        m.send_pkt(p)
    b = m.flush()
    logs = _read_logs(str(tmp_path))
    assert all(n in logs for n in TX_LOGS)
    cfg = make_cfg("qpsk")
    cfg.tx_amplitude = 1.0                       # ofdm_mod on its own: transmit_path's amplitude block is not in it
    N, CP = 512, 128
    oa, ob = (orc.tx(cfg, part, want_taps=True, want_ifft=True) for part in (pay[:3], pay[3:]))     # (one batch per flush)
    assert np.array_equal(a, oa[0]) and np.array_equal(b, ob[0])
    freq_o, ifft_o = np.concatenate([oa[1], ob[1]]), np.concatenate([oa[3], ob[3]])
    assert np.array_equal(logs["ofdm_preambles.dat"].reshape(-1, N), freq_o)
    assert np.array_equal(logs["ofdm_ifft_c.dat"].reshape(-1, N), ifft_o)
    assert np.array_equal(logs["ofdm_cp_adder_c.dat"].reshape(-1, N + CP)[:, CP:], ifft_o)
    assert len(logs["ofdm_mapper_c.dat"]) == (len(freq_o) - len(pay)) * N          # every symbol but the preambles
    # --- demodulator, one-shot calls: the second call appends to the first
    cfgc, payc, iq = _capture(orc, npkt=40, seed=9)
    d = ofdm.ofdm_demod(opt)
    first = d.work(iq[:200000])
    one = _read_logs(str(tmp_path))
    assert all(n in one for n in RX_LOGS)
    e = d.engine()
    assert np.array_equal(one["ofdm_receiver-chan_filt_c.dat"], e.tap(_abi.TAP_RX_CHAN_FILT))
    assert np.array_equal(one["ofdm_receiver-fft_out_c.dat"], e.tap(_abi.TAP_RX_FFT).reshape(-1))
    assert np.array_equal(one["ofdm_frame_sink_c.dat"], e.tap(_abi.TAP_RX_SINK).reshape(-1), equal_nan=True)
    d.work(iq[:200000])
    two = _read_logs(str(tmp_path))
    for n in RX_LOGS:
        assert len(two[n]) == 2 * len(one[n]) and np.array_equal(two[n][:len(one[n])], one[n], equal_nan=True), n
    # --- a fed stream: the files of the chunked run equal those of one call on the whole capture
    os.makedirs("whole")
    os.makedirs("chunked")
    monkeypatch.chdir(tmp_path / "whole")
    dw = ofdm.ofdm_demod(opt)
    want = dw.work(iq)
    whole = _read_logs(".")
    monkeypatch.chdir(tmp_path / "chunked")
    dc = ofdm.ofdm_demod(opt)
    got = []
    for k in range(0, len(iq), 150000):
        got += dc.feed(iq[k:k + 150000])
    got += dc.flush()
    assert got == want
    chunked = _read_logs(".")
    for n in RX_LOGS:
        assert len(chunked[n]) == len(whole[n]), n
        assert np.array_equal(chunked[n], whole[n], equal_nan=True), n
