"""Host-side mirror of the reference interface: names, defaults, option plumbing,
error behaviour (no GPU needed: nothing here creates an engine)."""
import optparse

import pytest

from helpers import make_cfg
from ofdm_uhd_amd import benchmark_ofdm_rx, benchmark_ofdm_tx, config, ofdm, options, receive_path, transmit_path


def _parser():
    p = optparse.OptionParser(option_class=options.eng_option, conflict_handler="resolve")
    return p, p.add_option_group("Expert")


def test_option_names_and_defaults_match_reference():
    p, e = _parser()
    transmit_path.transmit_path.add_options(p, e)
    receive_path.receive_path.add_options(p, e)
    ofdm.ofdm_mod.add_options(p, e)
    ofdm.ofdm_demod.add_options(p, e)
    o, _ = p.parse_args([])
    # ofdm.py:154-161, transmit_path.py:73-76
    assert (o.modulation, o.fft_length, o.occupied_tones, o.cp_length) == ("bpsk", 512, 200, 128)
    assert o.tx_amplitude == 0.25 and o.samples_per_symbol == 2 and o.verbose is False and o.log is False
    o, _ = p.parse_args(["-m", "qpsk", "--fft-length", "0x400", "--tx-amplitude", "100m", "-v"])
    assert o.modulation == "qpsk" and o.fft_length == 1024 and abs(o.tx_amplitude - 0.1) < 1e-12 and o.verbose
    d = options.default_options()
    assert (d.modulation, d.fft_length, d.occupied_tones, d.cp_length, d.tx_amplitude, d.snr, d.size) == \
        ("bpsk", 512, 200, 128, 0.25, 30, 1024)


def test_cfg_values():
    cfg = make_cfg("qpsk")
    assert (cfg.fft_length, cfg.occupied_tones, cfg.cp_length, cfg.arity, cfg.ntaps) == (512, 200, 128, 4, 155)
    assert cfg.phase_gain == 0.25 and cfg.freq_gain == 0.25 * 0.25 / 4          # ofdm.py:238-239
    assert abs(cfg.constellation[0].re - 0.707) < 1e-7 and abs(cfg.constellation[3].im + 0.707) < 1e-7
    assert cfg.tx_amplitude == 0.25 and cfg.sampler_timeout == 1000 and cfg.max_fft_shift_len == 4
    cfg = make_cfg("qam64", 4096, 2400, 1024)
    assert cfg.arity == 64 and cfg.ntaps == 103
    o = options.default_options(tx_amplitude=7.0)
    assert config.make_cfg(o).tx_amplitude == 1.0                               # clamp, transmit_path.py:56-62


def test_cfg_rejects_what_the_reference_rejects():
    with pytest.raises(ValueError):
        make_cfg("qpsk", 512, 600, 128)           # occupied > fft_length: mapper ctor throws
    with pytest.raises(ValueError):
        make_cfg("qpsk", 500, 200, 128)           # engine restriction: power-of-two FFT
    with pytest.raises(KeyError):
        make_cfg("qam1024")                        # mods[...] KeyError (ofdm.py:91-92)


def test_benchmark_tx_payload_construction(tmp_path):
    src = tmp_path / "tx1.txt"
    src.write_bytes(bytes(range(256)) * 20)
    o = options.default_options(size=1024, megabytes=1.0)
    with open(src, "rb") as f:
        pays = list(benchmark_ofdm_tx.build_payloads(o, f))
    # 20 garbage packets, then the file in chunks of size-2 (benchmark_ofdm_tx.py:111-117)
    assert len(pays) == 20 + 6
    assert pays[0] == b"\x00\x00\x00\x00This is Garbage data"
    assert pays[19][:4] == b"\x00\x13\x00\x00"
    assert pays[20][:4] == b"\x00\x14\x00\x00" and len(pays[20]) == 4 + 1022
    assert b"".join(p[4:] for p in pays[20:]) == src.read_bytes()


def test_benchmark_rx_accounting(tmp_path):
    out = open(tmp_path / "rx1.txt", "wb")
    acct = benchmark_ofdm_rx.rx_accounting(out, verbose=False)
    acct.rx_callback(True, b"\x00\x05\x00\x00garbage")      # pktno <= 19: counted, not written
    acct.rx_callback(True, b"\x00\x14\x00\x00hello ")
    acct.rx_callback(False, b"\x00\x15\x00\x00world")       # bad CRC still written (reference behaviour)
    acct.rx_callback(True, b"\x00\x16\x00\x01ignored")      # preamble field != 0: ignored entirely
    out.close()
    assert (acct.n_rcvd, acct.n_right) == (3, 2)
    assert (tmp_path / "rx1.txt").read_bytes() == b"hello world"
