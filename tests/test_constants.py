"""Constants of the product vs the fixtures extracted from the reference
(tests/golden/reference_constants.json, made by tests/golden/make_fixtures.py)."""
import math

import numpy as np

from ofdm_uhd_amd import config, firdes, ofdm_packet_utils, psk, qam


def test_known_symbols_match_reference(golden):
    ks = golden["known_symbols_4512_3"]
    assert len(config.known_symbols_4512_3) == 4512
    assert "".join("+" if v > 0 else "-" for v in config.known_symbols_4512_3) == ks


def test_whitening_mask_matches_reference(golden):
    mask = bytes.fromhex(golden["random_mask_hex"])
    assert bytes(ofdm_packet_utils.random_mask_tuple) == mask
    assert len(mask) == 4096 and list(mask[:8]) == [255, 63, 0, 16, 0, 12, 0, 5] and list(mask[-4:]) == [51, 51, 255, 63]


def test_constellations_bit_identical(golden):
    for k, v in golden["psk_gray_constellation"].items():
        assert [[c.real, c.imag] for c in psk.gray_constellation[int(k)]] == v
    for k, v in golden["psk_constellation"].items():
        assert [[c.real, c.imag] for c in psk.constellation[int(k)]] == v
    for k, v in golden["qam_constellation"].items():
        assert [[c.real, c.imag] for c in qam.constellation[int(k)]] == v


def test_mods_and_rotation(golden):
    assert config.MODS == golden["mods"]
    q = config.rotated_constellation("qpsk")
    rot = complex(*golden["qpsk_rot"])
    assert q == [p * rot for p in psk.gray_constellation[4]]
    assert abs(abs(q[0]) - 0.99985) < 1e-4          # 0.707+0.707j is NOT 1/sqrt(2) (ofdm.py:96)
    assert config.rotated_constellation("bpsk") == psk.gray_constellation[2]
    assert config.rotated_constellation("qam16") == qam.constellation[16]
    # mean powers quoted in SURVEY 8c
    for m, pw in ((4, 2.0), (16, 1.1111), (64, 0.85714), (256, 0.75556)):
        assert abs(np.mean(np.abs(np.array(qam.constellation[m])) ** 2) - pw) < 1e-4


def test_ksfreq_zeroes_odd_bins():
    for N, occ in ((512, 200), (2048, 1200), (4096, 2400)):
        zl = config.zeros_on_left(N, occ)
        assert zl == math.ceil((N - occ) / 2.0)
        ks = config.make_ksfreq(N, occ)
        pad = config.padded_preamble(N, occ)
        assert all(pad[i] == 0 for i in range(1, N, 2))
        assert all(abs(ks[i]) == 1 for i in range(0, occ, 2)) if zl % 2 == 0 else True
        # two identical halves in time <=> only even bins occupied
        t = np.fft.ifft(np.fft.ifftshift(np.array(pad, float)))
        assert np.allclose(t[:N // 2], t[N // 2:])


def test_carrier_map_sizing(orc):
    # SURVEY Appendix C
    m = config.carrier_map(200, 512)
    assert len(m) == 198 and m[0] == 156 and m[98] == 254 and m[99] == 257 and m[-1] == 355
    m = config.carrier_map(1200, 2048)
    assert len(m) == 1198 and m[0] == 424 and 1023 not in m and 1024 not in m and m[-1] == 1623
    m = config.carrier_map(2400, 4096)
    assert len(m) == 2398 and m[0] == 848 and 2047 not in m and 2048 not in m and m[-1] == 3247
    s = config.carrier_map(200, 200)
    assert len(s) == 198 and 99 not in s and 100 not in s
    for occ, cont in ((200, 512), (200, 200), (20, 64), (52, 64), (48, 48), (1200, 2048), (118, 256), (600, 600)):
        assert list(orc.carrier_map(occ, cont)) == config.carrier_map(occ, cont)


def test_channel_filter_taps():
    for N, occ, nt in ((512, 200, 155), (2048, 1200, 103), (4096, 2400, 103)):
        taps = config.channel_filter_taps(N, occ)
        assert len(taps) == nt
        assert abs(sum(taps) - 1.0) < 1e-12
        assert np.allclose(taps, taps[::-1])
    # pass band flat, stop band down (Hamming: ~53 dB)
    taps = np.array(config.channel_filter_taps(512, 200))
    H = np.abs(np.fft.fft(taps, 8192))
    f = np.arange(8192) / 8192.0
    assert np.all(np.abs(H[f < 0.19] - 1) < 5e-3)
    assert np.all(H[(f > 0.24) & (f < 0.5)] < 10 ** (-50 / 20))
