"""Runtime data-carrier maps: the hex-string rule of digital_ofdm_mapper_bcv /
digital_ofdm_frame_sink applied to maps the reference's sensor recorded.  CPU only: host
mirror vs oracle, and the oracle's own loopback."""
import json
import os

import numpy as np
import pytest

from helpers import loopback_stream, make_cfg, make_payloads
from ofdm_uhd_amd import config

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sensed():
    with open(os.path.join(HERE, "golden", "sense_blocks.json")) as f:
        return [b["carrier_map"] for b in json.load(f)["blocks"]]


def test_default_map_is_fe7f(orc, golden):
    assert config.carrier_map(200, 512) == config.carrier_map(200, 512, "FE7F") == orc.carrier_map(200, 512).tolist()
    assert len(config.carrier_map(200, 512)) == 198      # two DC carriers nulled


def test_host_and_oracle_agree_on_every_recorded_map(orc, sensed):
    for s in sensed:
        c = s[:50]                                         # sensing_and_tramsmitting.py:470
        for container in (512, 200):
            assert config.carrier_map(200, container, c) == orc.carrier_map(200, container, c).tolist()
        # bit k of the clipped map (MSB-first per digit) <-> occupied carrier k
        bits = [(int(ch, 16) >> (3 - j)) & 1 for ch in c for j in range(4)]
        assert config.carrier_map(200, 200, c) == [i for i, b in enumerate(bits) if b]
        assert config.carrier_map(200, 512, c) == [156 + i for i, b in enumerate(bits) if b]


def test_short_maps_grow_with_f(orc):
    # "FFFFFE7FFFFFF" (sensing_and_tramsmitting.py:461): 13 digits = 52 carriers.  Grown to 200
    # occupied tones it becomes 51 digits; the frame sink reads occ/4 + diff_left = 52 of them and
    # numbers the carriers from -diff_left: host mirror and oracle agree wherever the map is legal
    # and refuse the same ones.
    accepted = refused = 0
    for c in ("FFFFFE7FFFFFF", "0", "A5", "fe7f"):
        for occ, N in ((200, 512), (120, 256), (1200, 2048), (50, 64)):
            for container in (N, occ):
                try:
                    want = orc.carrier_map(occ, container, c).tolist()
                except ValueError:
                    with pytest.raises(ValueError):
                        config.carrier_map(occ, container, c)
                    refused += 1
                    continue
                assert config.carrier_map(occ, container, c) == want
                accepted += 1
    assert accepted >= 20 and refused >= 1
    assert len(config.carrier_map(204, 512, "FFFFFE7FFFFFF")) == 202


@pytest.mark.parametrize("mod,N,occ,CP", [("qpsk", 512, 180, 128), ("bpsk", 128, 100, 32), ("qpsk", 256, 204, 64)])
def test_occupied_tones_not_a_multiple_of_eight(orc, mod, N, occ, CP):
    """occ - 16 not a multiple of 8: the constructors add a partial nibble on each side.  The mapper centres the grown
    string in the fft_length bins, the frame sink numbers its carriers 4*i + j - diff_left inside the occupied block:
    the same carriers, so the loopback works (it must: the reference runs such sizes).  Host mirror, oracle and the
    float64 model agree."""
    import np_model as npm
    m, s = config.carrier_map(occ, N), config.carrier_map(occ, occ)
    zl = (N - occ + 1) // 2
    assert [c - zl for c in m] == s and s[0] == 0 and len(s) == occ - 2
    assert orc.carrier_map(occ, N).tolist() == m and orc.carrier_map(occ, occ).tolist() == s
    cfg = make_cfg(mod, N, occ, CP)
    pay = make_payloads(3, 200, seed=occ)
    x = loopback_stream(orc, cfg, pay, snr_db=30.0)
    r = orc.rx(cfg, x)
    assert r.packets == [(True, p) for p in pay]
    assert npm.rx(cfg, x)["packets"] == r.packets


def test_frame_sink_reads_occ_over_four_digits(orc):
    """digital_ofdm_frame_sink's constructor loop stops after occupied_tones/4 + diff_left digits: a longer string is
    clipped (sensing_and_tramsmitting.py:470 clips its 64-digit map to occ/4 itself), and a size whose grown string
    reaches past the occupied block (occ = 202: carrier 202) is refused where GNU Radio would read outside its
    input vector."""
    long = "F" * 10 + "FE7F" + "0" * 50                   # 64 digits
    assert config.carrier_map(200, 200, long) == config.carrier_map(200, 200, long[:50])
    assert orc.carrier_map(200, 200, long).tolist() == config.carrier_map(200, 200, long[:50])
    for bad in (202, 50):
        with pytest.raises(ValueError):
            config.carrier_map(bad, bad)
        with pytest.raises(ValueError):
            orc.carrier_map(bad, bad)
    with pytest.raises(ValueError):
        make_cfg("qpsk", 512, 202, 128)


def test_illegal_maps(orc, sensed):
    full = sensed[0]                                       # 64 digits, 241 ones > 200 occupied
    with pytest.raises(ValueError):
        config.carrier_map(200, 512, full)
    with pytest.raises(ValueError):
        orc.carrier_map(200, 512, full)
    with pytest.raises(ValueError):
        config.carrier_map(200, 512, "FE7G")
    with pytest.raises(ValueError):
        orc.carrier_map(200, 512, "FE7G")
    with pytest.raises(ValueError):
        make_cfg("qpsk", carriers=full)
    with pytest.raises(ValueError):
        make_cfg("qpsk", carriers="0" * 50)                # no data carrier left


@pytest.mark.parametrize("which", [0, 17, 36])
def test_oracle_loopback_on_sensed_map(orc, sensed, which):
    c = sensed[which][:50]
    cfg = make_cfg("qpsk", 512, 200, 128, carriers=c)
    pay = make_payloads(4, 500, seed=which)
    iq, freq, _ = orc.tx(cfg, pay, want_taps=True)
    used = np.flatnonzero(np.abs(freq[1]) > 0)
    assert used.tolist() == config.carrier_map(200, 512, c)
    # fewer carriers -> more symbols per packet than with the default map
    iq0 = orc.tx(make_cfg("qpsk", 512, 200, 128), pay)
    assert len(iq) > len(iq0)
    x = loopback_stream(orc, cfg, pay)
    r = orc.rx(cfg, x)
    assert r.packets == [(True, p) for p in pay]
    # a receiver left on the default map cannot read it
    r0 = orc.rx(make_cfg("qpsk", 512, 200, 128), x)
    assert [p for ok, p in r0.packets if ok] == []
