"""Host-side set-up of the OFDM modem: everything ``ofdm_mod.__init__``,
``ofdm_demod.__init__`` and ``ofdm_receiver.__init__`` compute before they build
their GNU Radio flow graphs (ofdm.py:63-101,204-247; ofdm_receiver.py~:69-98),
packed into the ``ofdm_cfg`` POD of include/ofdm_hip.h.
"""
import ctypes
import math

from . import _abi, firdes, psk, qam
from ._constants import KNOWN_SYMBOLS_COUNT, KNOWN_SYMBOLS_HEX, RANDOM_MASK_HEX

# ofdm.py:91 / ofdm.py:225
MODS = {"bpsk": 2, "qpsk": 4, "8psk": 8, "qam8": 8, "qam16": 16, "qam64": 64, "qam256": 256}


def _unpack_known_symbols():
    raw = bytes.fromhex(KNOWN_SYMBOLS_HEX)
    out = []
    for byte in raw:
        for b in range(7, -1, -1):
            out.append(1 if (byte >> b) & 1 else -1)
    assert len(out) == KNOWN_SYMBOLS_COUNT
    return out


# ofdm.py:310-325
known_symbols_4512_3 = _unpack_known_symbols()


def zeros_on_left(fft_length, occupied_tones):
    """ofdm.py:71"""
    return int(math.ceil((fft_length - occupied_tones) / 2.0))


def make_ksfreq(fft_length, occupied_tones):
    """Known symbol with every odd absolute bin zeroed, so the time-domain preamble
    is two identical halves (ofdm.py:71-77)."""
    zl = zeros_on_left(fft_length, occupied_tones)
    ksfreq = list(known_symbols_4512_3[0:occupied_tones])
    for i in range(len(ksfreq)):
        if (zl + i) & 1:
            ksfreq[i] = 0
    return ksfreq


def padded_preamble(fft_length, occupied_tones):
    """ofdm.py:83-87"""
    zl = zeros_on_left(fft_length, occupied_tones)
    padded = fft_length * [0, ]
    padded[zl:zl + occupied_tones] = make_ksfreq(fft_length, occupied_tones)
    return padded


def rotated_constellation(modulation):
    """ofdm.py:91-101: PSK tables are Gray coded, QAM tables as given; only "qpsk"
    is rotated, by the literal 0.707+0.707j."""
    if modulation not in MODS:
        raise KeyError(modulation)
    arity = MODS[modulation]
    rot = 1
    if modulation == "qpsk":
        rot = (0.707 + 0.707j)
    if modulation.find("psk") >= 0:
        return [pt * rot for pt in psk.gray_constellation[arity]]
    elif modulation.find("qam") >= 0:
        return [pt * rot for pt in qam.constellation[arity]]
    raise KeyError(modulation)


def channel_filter_taps(fft_length, occupied_tones):
    """ofdm_receiver.py~:69-75"""
    bw = (float(occupied_tones) / float(fft_length)) / 2.0
    tb = bw * 0.08
    return firdes.low_pass(1.0, 1.0, bw + tb, tb, firdes.WIN_HAMMING)


def carrier_map(occupied_tones, container, carriers="FE7F", sink=None):
    """Subcarrier map of digital_ofdm_mapper_bcv (container = fft_length, ofdm.py:106) or of
    digital_ofdm_frame_sink (``sink=True``; by default when container == occupied_tones, ofdm.py:240).
    Both grow the hex string with 'f' on both sides until it covers occupied_tones (a last partial
    nibble split ceil(diff/2) left, the rest right).  The mapper centres it in the fft_length bins in
    units of four carriers; the frame sink numbers carrier 4*i + j - diff_left inside the occupied
    block and reads only the first occupied_tones/4 + diff_left digits."""
    if sink is None:
        sink = container == occupied_tones
    s = carriers or "FE7F"
    for ch in s:
        if ch not in "0123456789abcdefABCDEF":
            raise ValueError("carrier map holds a non-hex digit: %r" % ch)
    diff = occupied_tones - 4 * len(s)
    while diff > 7:
        s = "f" + s + "f"
        diff -= 8
    dl = 0
    if diff > 0:
        dl = int(math.ceil(diff / 2.0))
        s = "0123456789abcdef"[(1 << dl) - 1] + s
        dr = diff - dl
        s = s + "0123456789abcdef"[0xF ^ ((1 << dr) - 1)]
    out = []
    if sink:
        for i in range(occupied_tones // 4 + dl):
            v = int(s[i], 16) if i < len(s) else 0
            for j in range(4):
                if (v >> (3 - j)) & 1:
                    out.append(4 * i + j - dl)
        limit = occupied_tones
    else:
        pad = int((container // 4 - len(s)) / 2)  # C integer division (truncates toward zero)
        for i, ch in enumerate(s):
            v = int(ch, 16)
            for j in range(4):
                if (v >> (3 - j)) & 1:
                    out.append(4 * (i + pad) + j)
        limit = container
    if len(out) > occupied_tones:
        raise ValueError("subcarriers allocated exceeds size of occupied carriers")
    if not out or min(out) < 0 or max(out) >= limit:
        raise ValueError("carrier map leaves no usable data carrier inside the container")
    return out


def make_cfg(options, pad_for_usrp=False, device_ptrs=False, device_id=0, pad_seed=0x0FD30000, carriers=None):
    """Build the engine configuration from an options object carrying the
    reference's attribute names (modulation, fft_length, occupied_tones, cp_length,
    tx_amplitude ...)."""
    N = int(options.fft_length)
    occ = int(options.occupied_tones)
    cp = int(options.cp_length)
    if N < 64 or N > _abi.OFDM_MAX_FFT or (N & (N - 1)):
        raise ValueError("fft_length must be a power of two in [64, %d]" % _abi.OFDM_MAX_FFT)
    if occ > N:
        # digital_ofdm_mapper_bcv ctor: std::invalid_argument
        raise ValueError("occupied carriers must be <= fft_length")
    if occ < 16:
        raise ValueError("occupied_tones must be >= 16")
    if cp < 1 or cp > N:
        raise ValueError("cp_length must be in [1, fft_length]")
    const = rotated_constellation(options.modulation)
    cfg = _abi.ofdm_cfg()
    cfg.struct_size = ctypes.sizeof(_abi.ofdm_cfg)
    cfg.device_id = int(device_id)
    cfg.flags = (_abi.OFDM_F_DEVICE_PTRS if device_ptrs else 0) | (_abi.OFDM_F_PAD_FOR_USRP if pad_for_usrp else 0)
    cfg.fft_length = N
    cfg.occupied_tones = occ
    cfg.cp_length = cp
    cfg.arity = len(const)
    for i, pt in enumerate(const):
        cfg.constellation[i].re = pt.real
        cfg.constellation[i].im = pt.imag
    for i, v in enumerate(make_ksfreq(N, occ)):
        cfg.known_symbol[i].re = float(v)
        cfg.known_symbol[i].im = 0.0
    ampl = getattr(options, "tx_amplitude", 0.25)
    cfg.tx_amplitude = max(0.0, min(float(ampl), 1))  # transmit_path.py:56-62
    phgain = 0.25                                     # ofdm.py:238
    cfg.phase_gain = phgain
    cfg.freq_gain = phgain * phgain / 4.0             # ofdm.py:239
    cfg.eq_gain = 0.05
    cfg.max_fft_shift_len = 4
    cfg.sampler_timeout = 1000
    cfg.peak_rise = 0.20
    cfg.peak_fall = 0.20
    cfg.peak_alpha = 0.001
    taps = channel_filter_taps(N, occ)
    if len(taps) > _abi.OFDM_MAX_TAPS:
        raise ValueError("channel filter needs %d taps (max %d)" % (len(taps), _abi.OFDM_MAX_TAPS))
    cfg.ntaps = len(taps)
    for i, t in enumerate(taps):
        cfg.taps[i] = t
    mask = bytes.fromhex(RANDOM_MASK_HEX)
    ctypes.memmove(cfg.whitening_mask, mask, len(mask))
    cfg.whitener_offset = 0
    cfg.pad_seed = int(pad_seed)
    if carriers is None:
        carriers = getattr(options, "carrier_map", None)
    # ofdm_receiver's SYNC selector (ofdm_receiver.py~:89): "pn" is the reference's hard-wired choice, "fixed" its
    # for-testing-only branch (:108-119: nsymbols = 18, freq_offset = 0.0); "ml" / "pnac" need blocks its tree lacks
    sync = getattr(options, "sync", "pn") or "pn"
    if sync not in ("pn", "fixed"):
        raise ValueError("SYNC must be 'pn' or 'fixed' (ofdm_sync_ml / ofdm_sync_pnac are not part of the reference's tree)")
    cfg.sync_mode = _abi.SYNC_FIXED if sync == "fixed" else _abi.SYNC_PN
    cfg.fixed_nsymbols = int(getattr(options, "sync_nsymbols", 18))
    cfg.fixed_freq_offset = float(getattr(options, "sync_freq_offset", 0.0))
    if cfg.sync_mode == _abi.SYNC_FIXED and cfg.fixed_nsymbols < 1:
        raise ValueError("sync_nsymbols must be >= 1")
    if carriers and len(carriers) > _abi.OFDM_MAX_CARRIER_HEX:
        raise ValueError("carrier map longer than %d hex digits" % _abi.OFDM_MAX_CARRIER_HEX)
    # raises ValueError exactly where the blocks' constructors would throw -- or would index outside their
    # vectors (e.g. occupied_tones = 202: the frame sink's loop reaches carrier 202)
    carrier_map(occ, N, carriers or "FE7F", sink=False)
    carrier_map(occ, occ, carriers or "FE7F", sink=True)
    if carriers:
        cfg.carrier_map = carriers.encode("ascii")
    return cfg


def make_sense_cfg(fft_size=256, tune_delay=24, dwell_delay=244, avg_msgs=10, skip_msgs=1, threshold=0.00010,
                   window=None):
    """ofdm_sense_cfg for the `sensor` flowgraph of predictive_sense.py:72-123 and its
    sense_loop (:150-222).  Defaults are the reference's: --fft-size 256 (:50),
    tune/dwell 1 ms / 10 ms at 6.25 MS/s in FFT frames (:113-116), 10 messages averaged
    (:159) with the 11th consumed unused (:174), threshold 1e-4 (:179)."""
    from . import window as _window
    n = int(fft_size)
    if n < 64 or n > _abi.OFDM_SENSE_MAX_FFT or (n & (n - 1)):
        raise ValueError("fft_size must be a power of two in [64, %d]" % _abi.OFDM_SENSE_MAX_FFT)
    if int(dwell_delay) < 1 or int(tune_delay) < 0 or int(avg_msgs) < 1 or int(skip_msgs) < 0:
        raise ValueError("dwell_delay >= 1, tune_delay >= 0, avg_msgs >= 1, skip_msgs >= 0")
    w = _window.blackmanharris(n) if window is None else list(window)
    if len(w) != n:
        # gr_fft_vcc_fftw::set_window refuses a window of the wrong length
        raise ValueError("window must have fft_size taps")
    sc = _abi.ofdm_sense_cfg()
    sc.struct_size = ctypes.sizeof(_abi.ofdm_sense_cfg)
    sc.fft_size = n
    sc.tune_delay = int(tune_delay)
    sc.dwell_delay = int(dwell_delay)
    sc.avg_msgs = int(avg_msgs)
    sc.skip_msgs = int(skip_msgs)
    sc.threshold = float(threshold)
    for i, v in enumerate(w):
        sc.window[i] = v
    return sc
