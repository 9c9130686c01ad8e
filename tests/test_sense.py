"""Spectrum sensor: host mirror + oracle against the reference's recorded
run logs (tests/golden/sense_blocks.json, made by tests/golden/make_sense_fixtures.py from
output.txt / output_with_detection.txt / crap.txt) and against an independent float64
NumPy model.  CPU only."""
import json
import os

import numpy as np
import pytest

from ofdm_uhd_amd import config, predictive_sense, window

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def blocks():
    with open(os.path.join(HERE, "golden", "sense_blocks.json")) as f:
        return json.load(f)


def py2_str(x):
    """str(float) of Python 2 (what `print p` wrote into the logs): 12 significant digits."""
    s = "%.12g" % x
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s


def test_fixture_shape(blocks):
    assert len(blocks["blocks"]) == 43
    assert set(blocks["freq_grids"]) == {"900000000.0", "920000000.0"}
    for b in blocks["blocks"]:
        assert len(b["power"]) == len(b["bit"]) == 256 and len(b["carrier_map"]) == 64


def test_hex_conv_recorded_blocks(blocks):
    """hex_conv(thrshold_inorder) == the 'Carrier map = ...' line, all 43 recorded blocks."""
    for b in blocks["blocks"]:
        assert predictive_sense.hex_conv(b["bit"]) == b["carrier_map"], (b["file"], b["line"])


def test_hex_conv_final_hex_conv_input(blocks):
    # final_hex_conv.py:37 feeds a '0'/'1' string; LSB-first nibbles: 0000 1111 1111 1111
    assert predictive_sense.hex_conv(blocks["final_hex_conv"]["bits"]) == "0FFF"
    assert predictive_sense.hex_conv([1, 0, 0, 0, 0, 1, 0, 0]) == "12"
    assert predictive_sense.hex_conv([1, 1, 1, 1, 1]) == "F"      # trailing partial group dropped
    assert predictive_sense.hex_conv([1, 1, 1]) == ""


def test_threshold_recorded_blocks(blocks):
    """bit = 0 if mean > 1e-4 else 1 (predictive_sense.py:179) on every recorded line."""
    thr = blocks["threshold"]
    n = 0
    for b in blocks["blocks"]:
        for p, bit in zip(b["power"], b["bit"]):
            assert (0 if float(p) > thr else 1) == bit
            n += 1
    assert n == 43 * 256


def test_sensed_freq_grid_recorded(blocks):
    for centre, grid in blocks["freq_grids"].items():
        got = predictive_sense.sensed_freq_grid(float(centre), blocks["samp_rate"], blocks["fft_size"])
        assert [py2_str(v) for v in got] == grid


def _msgs_from_block(b, S=256, nmsg=11):
    """Message bodies (FFT order) whose 10-message mean reproduces one recorded block."""
    p = np.array([float(v) for v in b["power"]], np.float64)
    fftorder = np.concatenate([p[S // 2:], p[:S // 2]])  # undo the half swap
    m = np.tile(fftorder.astype(np.float32), (nmsg, 1))
    m[10] = 1.0e3  # the message consumed by the else branch must not count (:174)
    return m


def test_oracle_decide_recorded_blocks(orc, blocks):
    sc = config.make_sense_cfg()
    msgs = np.concatenate([_msgs_from_block(b) for b in blocks["blocks"]])
    r = orc.sense_decide(sc, msgs)
    assert len(r["hex"]) == 43
    for d, b in enumerate(blocks["blocks"]):
        assert r["hex"][d] == b["carrier_map"], (b["file"], b["line"])
        assert r["bits"][d].tolist() == b["bit"]
        want = np.array([float(v) for v in b["power"]])
        assert np.allclose(r["mean"][d], want, rtol=2e-7, atol=0)


def test_sense_count(orc):
    sc = config.make_sense_cfg(256, 24, 244, 10, 1)
    per = (24 + 244) * 256
    from ofdm_uhd_amd import _abi
    import ctypes as C
    lib = _abi.load()
    for n in (0, per - 1, per, 11 * per - 1, 11 * per, 23 * per + 17):
        nm, nd = C.c_uint64(0), C.c_uint64(0)
        assert lib.ofdm_sense_count(C.byref(sc), n, C.byref(nm), C.byref(nd)) == 0
        assert (nm.value, nd.value) == orc.sense_count(sc, n) == (n // per, (n // per) // 11)
    bad = config.make_sense_cfg()
    bad.fft_size = 100
    assert lib.ofdm_sense_count(C.byref(bad), 1000, None, None) == _abi.OFDM_E_INVAL


def np_sense_msgs(sc, iq):
    """Independent float64 model of the sensor graph."""
    S = sc.fft_size
    w = np.array(sc.window[:S], np.float64)
    per = sc.tune_delay + sc.dwell_delay
    nm = (len(iq) // S) // per
    v = iq[:nm * per * S].astype(np.complex128).reshape(nm, per, S)[:, sc.tune_delay:, :]
    pw = np.abs(np.fft.fft(v * w, axis=2)) ** 2
    return pw.max(axis=1)


@pytest.mark.parametrize("S,tune,dwell", [(64, 0, 3), (256, 2, 5), (1024, 1, 4), (4096, 0, 2)])
def test_oracle_vs_numpy_model(orc, S, tune, dwell):
    rng = np.random.default_rng(S)
    sc = config.make_sense_cfg(S, tune, dwell, 3, 1, threshold=1e-4)
    n = (tune + dwell) * S * 9 + 5
    iq = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 1e-4
    k0 = S // 4 + 3
    iq += (0.05 * np.exp(2j * np.pi * k0 * np.arange(n) / S)).astype(np.complex64)
    r = orc.sense(sc, iq)
    ref = np_sense_msgs(sc, iq)
    assert r["msgs"].shape == ref.shape == (9, S)
    assert np.max(np.abs(r["msgs"] - ref)) <= 1e-5 * ref.max()
    assert len(r["hex"]) == 2
    # the tone's bin is occupied (0), far-away bins are free (1); in-order index = (k + S/2) % S
    for d in range(2):
        assert r["bits"][d][(k0 + S // 2) % S] == 0
        assert r["bits"][d][(k0 + S // 2 + S // 2) % S] == 1
        assert predictive_sense.hex_conv(r["bits"][d].tolist()) == r["hex"][d]
    # mean = float64 sum of the first 3 messages / 3, half-swapped
    m = (r["msgs"][0].astype(np.float64) + r["msgs"][1] + r["msgs"][2]) / 3.0
    assert np.array_equal(r["mean"][0], np.concatenate([m[S // 2:], m[:S // 2]]))


def test_oracle_noise_only_is_all_free(orc):
    rng = np.random.default_rng(5)
    sc = config.make_sense_cfg(256, 1, 3, 10, 1)
    n = 4 * 256 * 11
    iq = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1e-4).astype(np.complex64)
    r = orc.sense(sc, iq)
    assert r["hex"] == ["F" * 64]


def test_window_blackmanharris():
    w = window.blackmanharris(256)
    assert len(w) == 256
    assert abs(max(w) - 1.0) < 1e-3 and min(w) > 0 and min(w) < 1e-4 + 6e-5
    # sampled at (i + 0.5)/(n - 1): the mirror image of tap i is tap n - 2 - i
    assert np.allclose(w[:-1], w[:-1][::-1], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        config.make_sense_cfg(256, window=[1.0] * 255)
    with pytest.raises(ValueError):
        config.make_sense_cfg(100)


def test_sensor_options():
    tb = predictive_sense.sensor([])
    assert (tb.fft_size, tb.samp_rate, tb.tune_delay, tb.dwell_delay) == (256, 6.25e6, 24, 244)
    assert tb.min_center_freq == (1e7 + 1e8) / 2
    tb = predictive_sense.sensor(["-p", "905M", "-q", "895M", "-s", "512", "-d", "32", "--dwell-delay", "2m"])
    assert (tb.min_freq, tb.max_freq) == (895e6, 905e6)  # swapped (:66-68)
    assert tb.fft_size == 512 and tb.samp_rate == 3.125e6
    assert tb.dwell_delay == int(round(2e-3 * 3.125e6 / 512)) and tb.tune_delay == int(round(1e-3 * 3.125e6 / 512))
    assert tb.set_next_freq() == 900e6 and tb.set_next_freq() == 900e6  # freq_step = 0
    sc = tb.sense_cfg()
    assert (sc.fft_size, sc.avg_msgs, sc.skip_msgs, sc.threshold) == (512, 10, 1, 0.00010)
    assert predictive_sense.decimate_data([1, 3, 5, 7, 9], 2) == [2.0, 6.0]
