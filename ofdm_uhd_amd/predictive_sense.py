"""Host-side mirror of the reference's spectrum sensor (predictive_sense.py; the same
``sensor`` / ``sense_loop`` / ``hex_conv`` trio is pasted into sensing_and_tramsmitting*.py
and secondary_tx*.py).

The flowgraph  source -> stream_to_vector -> fft_vcc(window) -> complex_to_mag_squared ->
bin_statistics_f  and the averaging / threshold / reorder / hex tail run on the GPU
(csrc/sense.h) through ``Engine.sense``; this module keeps the reference's names, option
flags and return values.  The UHD source is replaced by a file / array source
(predictive_sense.py:92 already has the file variant); the named FIFOs of sense_loop
(:151-152) become optional file objects.
"""
import optparse
import time

import numpy as np

from . import config, iqio, window
from .options import eng_option


class parse_msg(object):
    """One bin_statistics_f message (predictive_sense.py:26-33): centre frequency,
    vector length and the float32 body."""

    def __init__(self, center_freq, data):
        self.center_freq = center_freq
        self.vlen = len(data)
        self.data = tuple(float(v) for v in data)
        self.raw_data = np.asarray(data, "<f4").tobytes()


def hex_conv(thrshold_inorder):
    """Vector of 1s and 0s -> upper-case hex string, four entries per character, FIRST
    entry = least significant bit; a trailing group shorter than four is dropped
    (predictive_sense.py:235-268).  Accepts the list of ints sense_loop builds or the
    '0'/'1' string final_hex_conv.py:37 feeds it."""
    abc = "0123456789ABCDEF"
    bits = [1 if (b == 1 or b == '1') else 0 for b in thrshold_inorder]
    out = []
    for i in range(0, len(bits) - 3, 4):
        out.append(abc[bits[i] + 2 * bits[i + 1] + 4 * bits[i + 2] + 8 * bits[i + 3]])
    return "".join(out)


def decimate_data(data, n):
    """Mean of each run of n entries (predictive_sense.py:228-232; the reference computes
    it and drops the result -- returned here)."""
    return [sum(data[i * n:n * (i + 1)]) / float(n) for i in range(len(data) // n)]


def sensed_freq_grid(center_freq, samp_rate, size):
    """Bin centre frequencies in the order sense_loop prints them (:188-205): starts
    size/2-1 bins below the centre and advances by repeated float addition."""
    freq_resolution = samp_rate / size
    p = center_freq - freq_resolution * ((size // 2) - 1)
    out = []
    for _ in range(size):
        out.append(p)
        p = p + freq_resolution
    return out


class sensor(object):
    """predictive_sense.sensor (:36-143) without the radio.

    ``argv`` carries the reference's flags (-p/-q start/stop, --tune-delay, --dwell-delay,
    -s/--fft-size, -d/--decim, -i/--input_file, -S/--sense-bins); ``source`` may be an
    ``iqio`` source object or an array instead of -i FILE."""

    def __init__(self, argv=None, source=None, engine=None, threshold=0.00010, avg_iterations=10):
        parser = optparse.OptionParser(option_class=eng_option)
        parser.add_option("-a", "--args", type="string", default="")
        parser.add_option("-p", "--start", type="eng_float", default=1e7)
        parser.add_option("-q", "--stop", type="eng_float", default=1e8)
        parser.add_option("", "--tune-delay", type="eng_float", default=1e-3, metavar="SECS")
        parser.add_option("", "--dwell-delay", type="eng_float", default=10e-3, metavar="SECS")
        parser.add_option("-g", "--gain", type="eng_float", default=None)
        parser.add_option("-s", "--fft-size", type="int", default=256)
        parser.add_option("-d", "--decim", type="intx", default=16)
        parser.add_option("-i", "--input_file", default="", metavar="FILE")
        parser.add_option("-S", "--sense-bins", type="int", default=64)
        (options, _args) = parser.parse_args(list(argv) if argv is not None else [])
        self.options = options
        self.min_freq, self.max_freq = options.start, options.stop
        if self.min_freq > self.max_freq:
            self.min_freq, self.max_freq = self.max_freq, self.min_freq  # :66-68
        self.fft_size = options.fft_size
        self.ofdm_bins = options.sense_bins
        self.mywindow = window.blackmanharris(self.fft_size)  # :73
        # the file branch's rate (:93); the USRP branch's `100**6/decim` (:86) is a typo for it
        self.samp_rate = 100e6 / options.decim
        if source is None and options.input_file:
            source = iqio.file_source(options.input_file, True)
        self.u = source
        self.freq_step = 0  # :95
        self.min_center_freq = (self.min_freq + self.max_freq) / 2
        nsteps = 10
        self.max_center_freq = self.min_center_freq + (nsteps * self.freq_step)
        self.next_freq = self.min_center_freq
        self.tune_delay = max(0, int(round(options.tune_delay * self.samp_rate / self.fft_size)))    # :113
        self.dwell_delay = max(1, int(round(options.dwell_delay * self.samp_rate / self.fft_size)))  # :115
        self.threshold = threshold
        self.avg_iterations = avg_iterations
        self._engine = engine

    # -- tuning (no radio: only the bookkeeping of :125-135) ------------------------
    def set_next_freq(self):
        target_freq = self.next_freq
        self.next_freq = self.next_freq + self.freq_step
        if self.next_freq >= self.max_center_freq:
            self.next_freq = self.min_center_freq
        return target_freq

    # -- engine ------------------------------------------------------------------------
    def sense_cfg(self):
        return config.make_sense_cfg(self.fft_size, self.tune_delay, self.dwell_delay, self.avg_iterations, 1,
                                     self.threshold, self.mywindow)

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            from .options import default_options
            self._engine = Engine(default_options())
        return self._engine

    def _samples(self, iq):
        if iq is not None:
            return np.ascontiguousarray(iq, np.complex64)
        if self.u is None:
            raise ValueError("sensor has no source: give -i FILE, source= or pass iq")
        if hasattr(self.u, "read_all"):
            return self.u.read_all()
        return np.ascontiguousarray(self.u, np.complex64)

    def run(self, iq=None):
        """Engine.sense over the whole source: dict with msgs / mean / bits / hex."""
        return self.engine().sense(self.sense_cfg(), self._samples(iq))

    def messages(self, iq=None):
        """The messages bin_statistics_f would post, in order (what tb.msgq delivers)."""
        res = self.run(iq)
        return [parse_msg(self.set_next_freq(), m) for m in res["msgs"]]


def sensor_init(argv=None, **kw):
    return sensor(argv, **kw)


def sense_loop(tb, iq=None, fifo=None, time_fifo=None, verbose=False):
    """predictive_sense.sense_loop (:150-222) over a finite stream: one entry per decision,
    each a dict with the reference's local names -- hexa_thr (what goes down the `fifo`),
    Time, thrshold_inorder, sensed_freq, moving_avg_data (ascending frequency, as printed by
    sensing_and_tramsmitting_first.py:231,238) and ofdm_center_freq.  ``fifo`` / ``time_fifo``
    are optional binary file objects standing in for the named pipes (:151-152,218-220)."""
    res = tb.run(iq)
    size = tb.fft_size
    out = []
    per = tb.avg_iterations + 1
    for d, hexa_thr in enumerate(res["hex"]):
        # set_next_freq is called once per message; the decision reads the centre of the
        # message that reached the else branch (:184)
        center = None
        for _ in range(per):
            center = tb.set_next_freq()
        Time = time.time()
        entry = {
            "hexa_thr": hexa_thr,
            "Time": Time,
            "thrshold_inorder": [int(b) for b in res["bits"][d]],
            "moving_avg_data": [float(v) for v in res["mean"][d]],
            "sensed_freq": sensed_freq_grid(center, tb.samp_rate, size),
            "ofdm_center_freq": center,
        }
        if verbose:
            print("hexa_thr= %s" % hexa_thr)
        if fifo is not None:
            fifo.write(hexa_thr.encode("ascii"))
        if time_fifo is not None:
            time_fifo.write(str(Time).encode("ascii"))
        out.append(entry)
    return out
