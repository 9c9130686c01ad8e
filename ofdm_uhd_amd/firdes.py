"""``gr.firdes.low_pass`` with a Hamming window -- the only filter design the
OFDM receiver uses (ofdm_receiver.py~:69-75).  Restates gr_firdes::low_pass of
GNU Radio 3.6.0: ntaps = int(53 * fs / (22 * width)) forced odd; windowed sinc;
taps normalised to the requested DC gain.  Designed in float64, handed to the
engine as float32 (``gr_fft_filter_ccc`` takes float taps).
"""
import math

WIN_HAMMING = 0
_MAX_ATTENUATION_HAMMING = 53.0


def compute_ntaps(sampling_freq, transition_width):
    ntaps = int(_MAX_ATTENUATION_HAMMING * sampling_freq / (22.0 * transition_width))
    if (ntaps & 1) == 0:
        ntaps += 1
    return ntaps


def low_pass(gain, sampling_freq, cutoff_freq, transition_width, window=WIN_HAMMING):
    if window != WIN_HAMMING:
        raise ValueError("only WIN_HAMMING is supported")
    if sampling_freq <= 0 or not (0 < cutoff_freq <= sampling_freq / 2) or transition_width <= 0:
        raise ValueError("firdes.low_pass: bad arguments")
    ntaps = compute_ntaps(sampling_freq, transition_width)
    M = (ntaps - 1) // 2
    fwT0 = 2 * math.pi * cutoff_freq / sampling_freq
    taps = [0.0] * ntaps
    for n in range(-M, M + 1):
        w = 0.54 - 0.46 * math.cos(2 * math.pi * (n + M) / (ntaps - 1))
        if n == 0:
            taps[n + M] = fwT0 / math.pi * w
        else:
            taps[n + M] = math.sin(n * fwT0) / (n * math.pi) * w
    fmax = taps[M]
    for n in range(1, M + 1):
        fmax += 2 * taps[n + M]
    g = gain / fmax
    return [t * g for t in taps]
