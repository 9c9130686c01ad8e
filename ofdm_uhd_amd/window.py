"""Window functions the reference takes from gnuradio.window (predictive_sense.py:73,
usrp_fft_save.py): GNU Radio 3.6 gnuradio/window.py is not in the reference tree, so
these follow its published definition [SURVEY A.13] -- PARITY UNPINNED for the tap
values (the sensing decisions downstream are pinned by the recorded run logs)."""
import math

# Blackman-Harris coefficient sets by side-lobe attenuation (dB)
_BH = {
    61: (0.44959, 0.49364, 0.05677, 0.0),
    67: (0.42323, 0.49755, 0.07922, 0.0),
    74: (0.40217, 0.49703, 0.09392, 0.00183),
    92: (0.35875, 0.48829, 0.14128, 0.01168),
}


def blackmanharris(fft_size, atten=92):
    """window.blackmanharris(fft_size): 4-term cosine sum sampled at (i + 0.5)/(fft_size - 1)."""
    a0, a1, a2, a3 = _BH[atten]
    out = []
    for i in range(fft_size):
        x = (i + 0.5) / (fft_size - 1)
        out.append(a0 - a1 * math.cos(2 * math.pi * x) + a2 * math.cos(4 * math.pi * x) - a3 * math.cos(6 * math.pi * x))
    return out


def rectangular(fft_size):
    return [1.0] * fft_size


def hamming(fft_size):
    return [0.54 - 0.46 * math.cos(2 * math.pi * i / (fft_size - 1)) for i in range(fft_size)]


def hanning(fft_size):
    return [0.5 - 0.5 * math.cos(2 * math.pi * i / (fft_size - 1)) for i in range(fft_size)]
