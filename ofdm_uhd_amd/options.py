"""Option objects with the reference's attribute names and defaults.

The reference gathers its flags with optparse + GNU Radio's ``eng_option``; each
layer contributes an ``add_options(normal, expert)`` static method
(ofdm.py:150-163,263-276; transmit_path.py:72-79; receive_path.py:45-48;
benchmark_ofdm_tx.py:62-70; benchmark_ofdm_rx.py:62-69).  ``default_options``
gives the same attribute set without a command line.
"""
import optparse


def _eng_float(option, opt, value):
    """GNU Radio eng_notation: 1k, 2.5M, 10m ... (eng_option's ``eng_float``)."""
    scale = {'T': 1e12, 'G': 1e9, 'M': 1e6, 'k': 1e3, 'm': 1e-3, 'u': 1e-6, 'n': 1e-9, 'p': 1e-12}
    try:
        if value and value[-1] in scale:
            return float(value[:-1]) * scale[value[-1]]
        return float(value)
    except ValueError:
        raise optparse.OptionValueError("option %s: invalid engineering notation value: %r" % (opt, value))


def _intx(option, opt, value):
    try:
        return int(value, 0)
    except ValueError:
        raise optparse.OptionValueError("option %s: invalid integer value: %r" % (opt, value))


class eng_option(optparse.Option):
    TYPES = optparse.Option.TYPES + ("eng_float", "intx")
    TYPE_CHECKER = dict(optparse.Option.TYPE_CHECKER)
    TYPE_CHECKER["eng_float"] = _eng_float
    TYPE_CHECKER["intx"] = _intx


def default_options(**overrides):
    """An options object carrying every hot-path flag at the reference's default."""
    v = optparse.Values()
    v.modulation = "bpsk"        # ofdm.py:154
    v.fft_length = 512           # ofdm.py:156
    v.occupied_tones = 200       # ofdm.py:158
    v.cp_length = 128            # ofdm.py:160
    v.tx_amplitude = 0.250       # transmit_path.py:73
    v.samples_per_symbol = 2     # transmit_path.py:75
    v.verbose = False
    v.log = False
    v.snr = 30                   # benchmark_ofdm_rx.py:64
    v.size = 1024                # benchmark_ofdm_tx.py:64
    v.megabytes = 1.0            # benchmark_ofdm_tx.py:65
    v.discontinuous = False
    v.from_file = None
    for k, val in overrides.items():
        setattr(v, k, val)
    return v
