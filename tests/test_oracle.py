"""Pins the C oracle (oracle/ofdm_oracle.c) against the independent float64 NumPy
model (tests/np_model.py), stage by stage, and checks the round trip.  The reference
has no golden vectors for this path (parity unpinned at the GNU Radio
boundary); what can be pinned from it is covered in test_constants / test_packet_utils.
"""
import numpy as np
import pytest

import np_model as npm
from helpers import MARGINAL, loopback_stream, make_cfg, make_payloads, marginal_capture
from ofdm_uhd_amd import _abi, config

CASES = [
    ("qpsk", 512, 200, 128, 1026, 4, 0.0),
    ("bpsk", 512, 200, 128, 300, 4, 0.05),
    ("qpsk", 512, 200, 128, 1026, 5, 0.3),     # first packet lost, a bogus header chains frames
    ("8psk", 256, 120, 64, 500, 3, 0.0),
    ("qam16", 2048, 1200, 512, 4091, 2, 0.0),
    ("qam64", 1024, 600, 256, 2000, 3, 0.1),
    ("qam256", 512, 200, 128, 777, 3, 0.0),
]
ALL_TAPS = sum(1 << t for t in (_abi.TAP_RX_CHAN_FILT, _abi.TAP_RX_METRIC, _abi.TAP_RX_FFT, _abi.TAP_RX_ACQ,
                                _abi.TAP_RX_SINK, _abi.TAP_RX_PACKETS))


@pytest.mark.parametrize("mod,N,occ,CP,plen,npkt,cfo", CASES)
def test_tx_matches_numpy_model(orc, mod, N, occ, CP, plen, npkt, cfo):
    cfg = make_cfg(mod, N, occ, CP)
    pay = make_payloads(npkt, plen)
    iq, freq, framed = orc.tx(cfg, pay, want_taps=True)
    f2 = npm.tx_freq(cfg, pay)
    assert freq.shape == f2.shape
    assert np.abs(freq - f2).max() < 1e-7          # table look-ups: float32 rounding of the constellation only
    t2 = npm.tx_time(cfg, f2)
    assert len(iq) == len(t2)
    assert np.abs(iq - t2).max() < 5e-7            # float32 FFT vs float64
    # symbol count: preamble + ceil(8*len / (carriers*nbits)) per packet
    ncar = len(config.carrier_map(occ, N))
    nb = npm.nbits_of(cfg)
    per = sum(1 + -(-8 * len(p) // (ncar * nb)) for p in framed)
    assert len(iq) == per * (N + CP)


@pytest.mark.parametrize("mod,N,occ,CP,plen,npkt,cfo", CASES)
def test_rx_matches_numpy_model(orc, mod, N, occ, CP, plen, npkt, cfo):
    cfg = make_cfg(mod, N, occ, CP)
    pay = make_payloads(npkt, plen)
    iq = loopback_stream(orc, cfg, pay, snr_db=30.0, cfo_bins=cfo)
    r = orc.rx(cfg, iq, ALL_TAPS)
    m = npm.rx(cfg, iq)
    assert np.abs(r.tap(_abi.TAP_RX_CHAN_FILT) - m["y"]).max() < 2e-6
    # metric: 1e-5 of (1 + M-bar).  Where a burst ends the window energy R collapses and M = |P|^2 / R^2 -- and with it
    # M-bar, up to ~30 -- amplifies the FFT filter's rounding (relative to the largest sample of its block): the bound
    # scales with M-bar there and stays at 1e-5 everywhere else (no global loosening)
    uo, un = r.tap(_abi.TAP_RX_METRIC).astype(np.float64), m["u"]
    assert np.all(np.abs(uo - un) <= 1e-5 * (1.0 + (un + 1.0)))
    assert list(r.tap(_abi.TAP_RX_PEAKS)) == list(m["peaks"])
    # angles: 5e-7 at the packets' flags; the false trigger where the burst ends takes the angle of a small,
    # ill-conditioned P (same remark as above): 1e-5
    da = np.abs(r.tap(_abi.TAP_RX_ANGLES) - m["angles"])
    assert da.max() < 1e-5 and np.sort(da)[:-1].max() < 2e-6
    # closed-form sampler (np_model.sampler_frames) == the automaton the oracle runs
    assert [tuple(x) for x in r.tap(_abi.TAP_RX_FRAMES)] == [tuple(x) for x in m["frames"]]
    F = r.tap(_abi.TAP_RX_FFT)
    F2 = np.array([s for _, s in m["fft"]])
    assert F.shape == F2.shape and np.abs(F - F2).max() < 1e-4 * max(1.0, np.abs(F2).max())
    # junk frames (false trigger at the burst end) have ill-conditioned equalisers: compare in relative terms
    A = r.tap(_abi.TAP_RX_ACQ)
    A2 = np.array([s for _, s in m["acq"]])
    assert A.shape == A2.shape
    assert np.all(np.abs(A - A2) <= 2e-4 * (1.0 + np.abs(A2)))
    S = r.tap(_abi.TAP_RX_SINK)
    S2 = np.array(m["sink"])
    nmap = len(config.carrier_map(occ, occ))
    assert S.shape[0] == S2.shape[0]
    assert np.all(np.abs(S[:, :nmap] - S2) <= 2e-4 * (1.0 + np.abs(S2)))
    assert r.packets == m["packets"]
    assert r.stats["packets"] == len(r.packets) and r.stats["crc_ok"] == sum(ok for ok, _ in r.packets)


@pytest.mark.parametrize("mod,N,occ,CP,plen,npkt,cfo", CASES)
def test_normative_detector_against_literal_recurrence(orc, mod, N, occ, CP, plen, npkt, cfo):
    """The peak detector's average has one normative evaluation (closed form over 2048-sample tiles: float32
    pre-selection metric outside the exact ranges, Q40 sums inside; oracle/ofdm_oracle.c peak_detect).  It must
    raise the flags gr_peak_detector_fb raises when run literally -- a float32 recurrence from the first sample --
    on every well-conditioned capture, and its float32 pre-selection must be the metric to float32 accuracy."""
    cfg = make_cfg(mod, N, occ, CP)
    pay = make_payloads(npkt, plen)
    iq = loopback_stream(orc, cfg, pay, snr_db=30.0, cfo_bins=cfo)
    r = orc.rx(cfg, iq, (1 << _abi.TAP_RX_METRIC) | (1 << _abi.TAP_RX_PRESEL))
    assert r.tap(_abi.TAP_RX_PEAKS).tolist() == r.tap(orc.TAP_PEAKS_GR).tolist()
    assert r.presel_miss == 0          # no sample above the candidate threshold escaped the pre-selection
    u, u32 = r.tap(_abi.TAP_RX_METRIC).astype(np.float64), r.tap(_abi.TAP_RX_PRESEL).astype(np.float64)
    # float32 running sums anchored per tile: ~1e-6 in general; where a burst ends the window energy collapses by
    # the SNR and the sums lose the small R to cancellation (M-bar up to ~30 there): relative 1e-3
    assert np.median(np.abs(u32 - u)) < 1e-6
    assert np.all(np.abs(u32 - u) <= 1e-3 * (1.0 + np.abs(u + 1.0)))
    rg = r.tap(orc.TAP_RANGES)
    sel = rg[:, 1] >= 0
    assert sel.any() and not sel.all() or len(rg) < 6      # the exact evaluation covers a fraction of the tiles
    # every flag lies inside the range of its tile
    for p in r.tap(_abi.TAP_RX_PEAKS):
        g, k = int(p) // 2048, int(p) % 2048
        assert rg[g, 0] <= k <= rg[g, 1]


@pytest.mark.parametrize("which", range(len(MARGINAL)))
def test_marginal_captures_split_the_two_evaluations(orc, which):
    """The round-2 soak captures on which engine and oracle disagreed (non-default thresholds: u drifts across
    avg * rise within the average's rounding noise): the normative evaluation and the literal recurrence differ by
    exactly the recorded run.  (The engine side of these: tests/test_gpu_parity.py::test_marginal_detector_cases.)"""
    d = MARGINAL[which]
    cfg, x = marginal_capture(orc, d)
    r = orc.rx(cfg, x, 0)
    flags, lit = r.tap(_abi.TAP_RX_PEAKS).tolist(), r.tap(orc.TAP_PEAKS_GR).tolist()
    assert (len(flags), len(lit)) == (d["flags"], d["literal"])
    assert sorted(set(flags) ^ set(lit)) == [d["at"]]
    assert r.presel_miss == 0


def test_loopback_recovers_packets(orc):
    """BPSK/QPSK at the reference's default 30 dB: every packet comes back.  The QAM sizes do not get
    there at ANY noise level: the reference flags the END of the Schmidl-Cox plateau, which the first data
    symbol's content drags by tens of samples, and 16/64-QAM cannot absorb the resulting ISI (about 5 %
    of the packets, the same ones at 30, 35 and 40 dB).  That is behaviour to reproduce, not to fix."""
    for mod, plen, snr, floor in (("bpsk", 64, 30.0, 1.0), ("qpsk", 1026, 30.0, 1.0), ("qam16", 1500, 35.0, 0.8),
                                  ("qam64", 2000, 40.0, 0.7)):
        cfg = make_cfg(mod)
        npkt = 6 if floor == 1.0 else 40
        pay = make_payloads(npkt, plen, seed=7, variant="ref")
        iq = loopback_stream(orc, cfg, pay, snr_db=snr)
        r = orc.rx(cfg, iq)
        good = [p for ok, p in r.packets if ok]
        assert all(p in pay for p in good)
        assert len(good) >= floor * npkt, (mod, len(good))
        if floor == 1.0:
            assert good == pay
            # GR's metric spikes when the burst ends: one extra flag, no extra packet
            assert r.stats["peaks"] == len(pay) + 1 and r.stats["packets"] == len(pay)


def test_ragged_and_empty_payloads(orc):
    cfg = make_cfg("qpsk")
    pay = make_payloads(7, [0, 1, 4, 5, 100, 1026, 4091], seed=11)
    iq = loopback_stream(orc, cfg, pay)
    r = orc.rx(cfg, iq)
    assert [p for ok, p in r.packets] == pay
    # a 0..3 byte payload still passes the CRC (message is >= 4 bytes)
    assert all(ok for ok, _ in r.packets)
    # no packets, no samples
    assert orc.rx(cfg, np.zeros(0, np.complex64)).packets == []
    noise = np.zeros(20000, np.complex64)
    orc.channel(noise, sigma=0.01)
    assert orc.rx(cfg, noise).packets == []


def test_sampler_timeout_path(orc):
    """A packet, more than 1001 symbol times of noise (sampler time-out -> NO_SIG grid), another
    packet: the closed form used on the GPU must agree with the automaton."""
    cfg = make_cfg("qpsk", 64, 48, 16)
    N, CP = 64, 16
    pay = make_payloads(2, 60, seed=6)
    a = orc.tx(cfg, pay[:1], lead=2 * N, tail=0)
    gap = np.zeros(1100 * (N + CP) + 37, np.complex64)
    b = orc.tx(cfg, pay[1:], lead=0, tail=(N + CP) + 2 * N)
    iq = np.concatenate([a, gap, b])
    sigma = float(np.sqrt(np.mean(np.abs(a[2 * N:]) ** 2) / 1000.0))
    for seed in range(6):     # the closed form must hold for every noise realisation, and the time-out must fire in each
        x = iq.copy()
        orc.channel(x, sigma=sigma, seed=seed)
        r = orc.rx(cfg, x, 1 << _abi.TAP_RX_METRIC)
        peaks = r.tap(_abi.TAP_RX_PEAKS)
        frames = npm.sampler_frames(peaks, len(x), N, CP, cfg.sampler_timeout)
        assert [tuple(f) for f in r.tap(_abi.TAP_RX_FRAMES)] == frames
        assert max(k for _, k in frames) == cfg.sampler_timeout + 1
        if seed == 0:         # (this realisation also brings both packets back: N = 64 at 30 dB loses one in the others)
            assert [p for ok, p in r.packets if ok] == pay


def test_pad_symbols_are_counter_based(orc):
    # the mapper's rand()%arity fill is replaced by a hash of (seed, packet, slot): deterministic
    for args in ((1, 2, 3, 4), (0x0FD30000, 0, 4140, 64), (2 ** 63, 65535, 10 ** 6, 256)):
        assert orc.lib().orc_pad_symbol(*args) == npm.pad_symbol(*args)


def test_channel_generator_known_answers(orc):
    """The synthetic channel draws from Philox-2x32-7; these are the Random123 known-answer vectors of the
    seven-round and of the ten-round generator (key folded from seed with stream 0: key = seed_lo ^ seed_hi)."""
    import ctypes as C
    out = (C.c_uint32 * 2)()

    def ph(c0, c1, k, rounds=None):
        if rounds is None:
            orc.lib().orc_philox(C.c_uint64(k), C.c_uint64(0), C.c_uint64((c1 << 32) | c0), out)
        else:
            orc.lib().orc_philox_r(C.c_uint64(k), C.c_uint64(0), C.c_uint64((c1 << 32) | c0), rounds, out)
        return (out[0], out[1])
    for rounds in (None, 7):     # the channel's generator IS the seven-round one
        assert ph(0, 0, 0, rounds) == (0x257a3673, 0xcd26be2a)
        assert ph(0xffffffff, 0xffffffff, 0xffffffff, rounds) == (0xab302c4d, 0x3dc9d239)
        assert ph(0x243f6a88, 0x85a308d3, 0x13198a2e, rounds) == (0xbedbbe6b, 0xe4c770b3)
    assert ph(0, 0, 0, 10) == (0xff1dae59, 0x6cd10df2)
    assert ph(0xffffffff, 0xffffffff, 0xffffffff, 10) == (0x2c3f628b, 0xab4fd7ad)
    assert ph(0x243f6a88, 0x85a308d3, 0x13198a2e, 10) == (0xdd7ce038, 0xf62a4c12)
    # unit-variance circular noise, independent per stream
    a = np.zeros(200000, np.complex64)
    b = np.zeros(200000, np.complex64)
    orc.channel(a, sigma=1.0, stream_id=0)
    orc.channel(b, sigma=1.0, stream_id=1)
    assert abs(np.mean(np.abs(a) ** 2) - 1.0) < 0.01 and abs(np.mean(a)) < 0.01
    assert abs(np.mean(a * np.conj(b))) < 0.01
    assert abs(np.mean(a.real * a.imag)) < 0.01


def test_normative_fft_matches_numpy(orc):
    """The oracle's transform (the schedule the engine mirrors bit for bit) is a DFT: float32 rounding away from
    numpy's float64 result, for every length the engine builds (OFDM symbol, channel filter, sensor)."""
    rng = np.random.default_rng(11)
    for n in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        X = np.fft.fft(x.astype(np.complex128))
        scale = np.abs(X).max()
        assert np.abs(orc.fft(x) - X).max() < 4e-7 * scale, n
        assert np.abs(orc.fft(x, inverse=True) - np.fft.ifft(x.astype(np.complex128)) * n).max() < 4e-7 * scale, n
        # round trip: unnormalised both ways
        assert np.abs(orc.fft(orc.fft(x), inverse=True) / n - x).max() < 1e-6 * np.abs(x).max()


def test_deterministic_sincos_accuracy(orc):
    """The bit-reproducible sin / cos both sides evaluate instead of libm / ocml are accurate to 2 ulp
    (float32, over the arguments the receiver produces) and 2e-16 (float64 NCO phasor)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-7000, 7000, 200000), rng.uniform(0, 2 * np.pi, 200000),
                        np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 6283.1855])]).astype(np.float32)
    s, c = orc.sincosf(x)
    xs = x.astype(np.float64)
    ulp = np.spacing(np.float32(1.0))
    assert np.abs(s - np.sin(xs)).max() < 2 * ulp and np.abs(c - np.cos(xs)).max() < 2 * ulp
    s0, c0 = orc.sincosf(np.array([-0.0], np.float32))
    assert c0[0] == 1.0 and s0[0] == 0.0 and not np.signbit(s0[0])      # what the engine's coarse == 0 shortcut assumes
    ph = np.concatenate([rng.uniform(-1e6, 1e6, 200000), rng.uniform(-np.pi, np.pi, 200000)])
    z = orc.expj(ph)
    red = ph - 2 * np.pi * np.floor(ph / (2 * np.pi) + 0.5)
    assert np.abs(z - np.exp(1j * red)).max() < 4e-16


def test_sync_fixed_oracle_loopback(orc):
    """SYNC = "fixed" in the oracle: flags at (N+CP)-1 + k*nsymbols*(N+CP), chan_filt = x, constant NCO input."""
    from ofdm_uhd_amd import options
    N, CP = 256, 64
    pay = make_payloads(5, 300, seed=4)
    nsym = len(orc.tx(make_cfg("qpsk", N, 120, CP), pay[:1])) // (N + CP)
    for fo_bins in (0.0, -0.3):
        opt = options.default_options(modulation="qpsk", fft_length=N, occupied_tones=120, cp_length=CP, sync="fixed",
                                      sync_nsymbols=nsym, sync_freq_offset=float(np.pi * fo_bins))
        cfg = config.make_cfg(opt)
        x = loopback_stream(orc, cfg, pay, snr_db=30.0, cfo_bins=fo_bins, lead=0, tail=100)
        r = orc.rx(cfg, x, (1 << _abi.TAP_RX_CHAN_FILT) | (1 << _abi.TAP_RX_NCO))
        assert np.array_equal(r.tap(_abi.TAP_RX_CHAN_FILT), x)
        assert list(r.tap(_abi.TAP_RX_PEAKS)) == [N + CP - 1 + k * nsym * (N + CP) for k in range(5)]
        assert [p for ok, p in r.packets if ok] == pay
        # gr_frequency_modulator_fc driven by a constant: phi[n] = -2/N * f * (n+1)
        n = np.arange(len(x))
        want = np.exp(1j * (-2.0 / N) * np.float32(np.pi * fo_bins) * (n + 1))
        assert np.abs(r.tap(_abi.TAP_RX_NCO) - want).max() < 2e-6
    with pytest.raises(ValueError):
        config.make_cfg(options.default_options(sync="ml"))
