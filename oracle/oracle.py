"""ctypes wrapper around oracle/libofdm_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It shares the ``ofdm_cfg`` layout with the product ABI, so tests
build one configuration and hand it to both sides.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ofdm_uhd_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libofdm_oracle.so")
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [vp, C.c_uint64]
        L.orc_atan2f.restype = C.c_float
        L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_sincosf_vec.restype = None
        L.orc_sincosf_vec.argtypes = [vp, C.c_uint64, vp, vp]
        L.orc_expj_vec.restype = None
        L.orc_expj_vec.argtypes = [vp, C.c_uint64, vp, vp]
        L.orc_fft.argtypes = [vp, C.c_int, C.c_int]
        L.orc_filter_fft_len.argtypes = [C.c_int]
        L.orc_framed_len.argtypes = [C.POINTER(_abi.ofdm_cfg), C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_make_packet.argtypes = [C.POINTER(_abi.ofdm_cfg), vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]
        L.orc_unmake_packet.argtypes = [C.POINTER(_abi.ofdm_cfg), vp, C.c_uint32, vp, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_int)]
        L.orc_carrier_map.argtypes = [C.c_int, C.c_int, C.c_char_p, vp, C.c_int]
        L.orc_tx_data_symbols.restype = C.c_uint32
        L.orc_tx_data_symbols.argtypes = [C.POINTER(_abi.ofdm_cfg), C.c_uint32, C.c_int]
        L.orc_pad_symbol.restype = C.c_uint32
        L.orc_pad_symbol.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_tx.argtypes = [C.POINTER(_abi.ofdm_cfg), vp, vp, vp, C.c_int, C.c_uint64, vp, C.c_uint64,
                             C.POINTER(C.c_uint64), vp, vp, vp]
        L.orc_tx_ex.argtypes = [C.POINTER(_abi.ofdm_cfg), vp, vp, vp, C.c_int, C.c_uint64, vp, C.c_uint64,
                                C.POINTER(C.c_uint64), vp, vp, vp, vp]
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp]
        L.orc_philox.restype = None
        L.orc_philox_r.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, vp]
        L.orc_philox_r.restype = None
        L.orc_channel.argtypes = [vp, C.c_uint64, C.POINTER(_abi.ofdm_chan), C.c_uint64]
        L.orc_rx.restype = vp
        L.orc_rx.argtypes = [C.POINTER(_abi.ofdm_cfg), vp, C.c_uint64, C.c_uint32]
        L.orc_rx_tap.restype = C.c_uint64
        L.orc_rx_tap.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.orc_rx_npackets.argtypes = [vp]
        L.orc_rx_payload_bytes.restype = C.c_uint64
        L.orc_rx_payload_bytes.argtypes = [vp]
        L.orc_rx_packets.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_int]
        L.orc_rx_stats.argtypes = [vp, C.POINTER(_abi.ofdm_stats)]
        L.orc_rx_stats.restype = None
        L.orc_rx_free.argtypes = [vp]
        L.orc_rx_presel_miss.restype = C.c_uint64
        L.orc_rx_presel_miss.argtypes = [vp]
        SC = C.POINTER(_abi.ofdm_sense_cfg)
        L.orc_sense_count.argtypes = [SC, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_sense.argtypes = [SC, vp, C.c_uint64, vp, vp, vp, vp]
        L.orc_sense_decide.argtypes = [SC, vp, C.c_uint64, vp, vp, vp]
        L.orc_rx_free.restype = None
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def sincosf(x):
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_sincosf_vec(_ptr(x), x.size, _ptr(s), _ptr(c))
    return s, c


def expj(ph):
    ph = np.ascontiguousarray(ph, np.float64)
    re, im = np.empty_like(ph), np.empty_like(ph)
    lib().orc_expj_vec(_ptr(ph), ph.size, _ptr(re), _ptr(im))
    return re + 1j * im


def fft(x, inverse=False):
    """The normative transform (unnormalised either way), in place on a copy."""
    y = np.ascontiguousarray(x, np.complex64).copy()
    rc = lib().orc_fft(_ptr(y), y.size, 1 if inverse else 0)
    if rc:
        raise ValueError("orc_fft: %d" % rc)
    return y


def filter_fft_len(ntaps):
    return int(lib().orc_filter_fft_len(int(ntaps)))


def crc32(data):
    b = np.frombuffer(bytes(data), np.uint8)
    return int(lib().orc_crc32(_ptr(b) if len(b) else None, len(b)))


def make_packet(cfg, payload):
    payload = bytes(payload)
    out = np.zeros(len(payload) + 64, np.uint8)
    n = C.c_uint32(0)
    pin = np.frombuffer(payload, np.uint8) if payload else np.zeros(1, np.uint8)
    rc = lib().orc_make_packet(C.byref(cfg), _ptr(pin), len(payload), _ptr(out), C.byref(n))
    if rc:
        raise ValueError("orc_make_packet rc=%d" % rc)
    return out[:n.value].tobytes()


def unmake_packet(cfg, msg):
    msg = bytes(msg)
    out = np.zeros(max(len(msg), 1), np.uint8)
    n = C.c_uint32(0)
    ok = C.c_int(0)
    min_ = np.frombuffer(msg, np.uint8) if msg else np.zeros(1, np.uint8)
    rc = lib().orc_unmake_packet(C.byref(cfg), _ptr(min_), len(msg), _ptr(out), C.byref(n), C.byref(ok))
    if rc:
        raise ValueError("orc_unmake_packet rc=%d" % rc)
    return bool(ok.value), out[:n.value].tobytes()


def carrier_map(occ, container, carriers="FE7F"):
    m = np.zeros(_abi.OFDM_MAX_FFT, np.int32)
    n = lib().orc_carrier_map(occ, container, carriers.encode("ascii"), _ptr(m), len(m))
    if n < 0:
        raise ValueError("orc_carrier_map rc=%d" % n)
    return m[:n].copy()


def pack_payloads(payloads):
    lens = np.array([len(p) for p in payloads], np.uint32)
    offs = np.zeros(len(payloads), np.uint64)
    if len(payloads):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bytes(p) for p in payloads), np.uint8) if lens.sum() else np.zeros(1, np.uint8)
    return np.ascontiguousarray(blob), offs, lens


def tx(cfg, payloads, lead=0, tail=0, want_taps=False, want_ifft=False):
    """Returns iq (complex64, lead + symbols + tail, zeros outside the symbols) and
    optionally (freq-domain symbols [nsym, N], framed packets list[, transform output [nsym, N]])."""
    L = lib()
    blob, offs, lens = pack_payloads(payloads)
    N, CP = cfg.fft_length, cfg.cp_length
    ncar = len(carrier_map(cfg.occupied_tones, N, cfg.carrier_map.decode('ascii') or 'FE7F'))
    nsym = 0
    flens = []
    for ln in lens:
        fl = C.c_uint32(0)
        rc = L.orc_framed_len(C.byref(cfg), int(ln), C.byref(fl))
        if rc:
            raise ValueError("len(payload) must be in [0, %d]" % (_abi.OFDM_MASK_LEN - 4))
        flens.append(fl.value)
        nsym += 1 + L.orc_tx_data_symbols(C.byref(cfg), fl.value, ncar)
    total = lead + nsym * (N + CP) + tail
    iq = np.zeros(total, np.complex64)
    ns = C.c_uint64(0)
    freq = np.zeros((nsym, N), np.complex64) if want_taps else None
    framed = np.zeros(sum(flens) + 1, np.uint8) if want_taps else None
    foff = np.zeros(len(payloads) + 1, np.uint64) if want_taps else None
    ifft = np.zeros((nsym, N), np.complex64) if want_ifft else None
    rc = L.orc_tx_ex(C.byref(cfg), _ptr(blob), _ptr(offs), _ptr(lens), len(payloads), lead, _ptr(iq), total,
                     C.byref(ns), _ptr(freq) if want_taps else None, _ptr(framed) if want_taps else None,
                     _ptr(foff) if want_taps else None, _ptr(ifft) if want_ifft else None)
    if rc:
        raise ValueError("orc_tx rc=%d" % rc)
    assert ns.value == lead + nsym * (N + CP)
    if want_taps:
        pk = [framed[int(foff[i]):int(foff[i + 1])].tobytes() for i in range(len(payloads))]
        return (iq, freq, pk, ifft) if want_ifft else (iq, freq, pk)
    return (iq, ifft) if want_ifft else iq


def channel(iq, sigma=0.0, cfo=0.0, seed=0xC0FFEE, stream_id=0, index0=0):
    """In-place channel on a complex64 array."""
    ch = _abi.ofdm_chan(sigma=sigma, cfo=cfo, seed=seed, stream_id=stream_id, lead_samples=0, tail_samples=0)
    assert iq.dtype == np.complex64 and iq.flags.c_contiguous
    lib().orc_channel(_ptr(iq), len(iq), C.byref(ch), index0)
    return iq


_TAP_DTYPES = {
    _abi.TAP_RX_CHAN_FILT: np.complex64, _abi.TAP_RX_METRIC: np.float32, _abi.TAP_RX_PEAKS: np.uint64,
    _abi.TAP_RX_ANGLES: np.float32, _abi.TAP_RX_FRAMES: np.uint64, _abi.TAP_RX_FFT: np.complex64,
    _abi.TAP_RX_ACQ: np.complex64, _abi.TAP_RX_SINK: np.complex64, _abi.TAP_RX_PACKETS: np.uint8,
    _abi.TAP_RX_SAMPLER: np.complex64, _abi.TAP_RX_SIGMIX: np.complex64, _abi.TAP_RX_NCO: np.complex64,
    _abi.TAP_RX_PRESEL: np.float32,
}
# oracle-only taps: the literal float32-recurrence detector's flags (cross-check of the normative evaluation) and the
# exact-evaluation range of every 2048-sample tile
TAP_PEAKS_GR, TAP_RANGES = 100, 102
_TAP_DTYPES[TAP_PEAKS_GR] = np.uint64
_TAP_DTYPES[TAP_RANGES] = np.int32


class RxResult(object):
    def __init__(self, cfg, iq, tap_mask=0):
        iq = np.ascontiguousarray(iq, np.complex64)
        self._cfg = cfg
        self._h = lib().orc_rx(C.byref(cfg), _ptr(iq) if len(iq) else None, len(iq), tap_mask)
        st = _abi.ofdm_stats()
        lib().orc_rx_stats(self._h, C.byref(st))
        self.stats = st.as_dict()
        # samples above the candidate threshold that the float32 pre-selection left outside every range (expected: 0)
        self.presel_miss = int(lib().orc_rx_presel_miss(self._h))
        np_ = lib().orc_rx_npackets(self._h)
        nb = lib().orc_rx_payload_bytes(self._h)
        pay = np.zeros(max(nb, 1), np.uint8)
        off = np.zeros(np_ + 1, np.uint64)
        ln = np.zeros(max(np_, 1), np.uint32)
        ok = np.zeros(max(np_, 1), np.uint8)
        rc = lib().orc_rx_packets(self._h, _ptr(pay), len(pay), _ptr(off), _ptr(ln), _ptr(ok), np_)
        assert rc == np_, rc
        self.packets = [(bool(ok[i]), pay[int(off[i]):int(off[i]) + int(ln[i])].tobytes()) for i in range(np_)]

    def tap(self, tap):
        nb = lib().orc_rx_tap(self._h, tap, None, 0)
        dt = np.dtype(_TAP_DTYPES[tap])
        out = np.zeros(nb // dt.itemsize, dt)
        if nb:
            lib().orc_rx_tap(self._h, tap, _ptr(out), nb)
        if tap in (_abi.TAP_RX_FRAMES, TAP_RANGES):
            out = out.reshape(-1, 2)
        elif tap in (_abi.TAP_RX_FFT, _abi.TAP_RX_SAMPLER):
            out = out.reshape(-1, self._cfg.fft_length)
        elif tap in (_abi.TAP_RX_ACQ, _abi.TAP_RX_SINK):
            out = out.reshape(-1, self._cfg.occupied_tones)
        return out

    def close(self):
        if self._h:
            lib().orc_rx_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def rx(cfg, iq, tap_mask=0):
    return RxResult(cfg, iq, tap_mask)


def sense_count(sc, nsamples):
    nm, nd = C.c_uint64(0), C.c_uint64(0)
    rc = lib().orc_sense_count(C.byref(sc), int(nsamples), C.byref(nm), C.byref(nd))
    if rc:
        raise ValueError("orc_sense_count: %d" % rc)
    return nm.value, nd.value


def _sense_out(sc, nm, nd):
    S = sc.fft_size
    return (np.zeros((max(nd, 1), S), np.float64), np.zeros((max(nd, 1), S), np.uint8),
            np.zeros((max(nd, 1), S // 4), np.uint8))


def sense(sc, iq):
    """predictive_sense.py sensor graph + sense_loop: same dict as Engine.sense."""
    iq = np.ascontiguousarray(iq, np.complex64)
    nm, nd = sense_count(sc, len(iq))
    msgs = np.zeros((max(nm, 1), sc.fft_size), np.float32)
    mean, bits, hexs = _sense_out(sc, nm, nd)
    rc = lib().orc_sense(C.byref(sc), _ptr(iq), len(iq), _ptr(msgs), _ptr(mean), _ptr(bits), _ptr(hexs))
    if rc:
        raise ValueError("orc_sense: %d" % rc)
    return {"msgs": msgs[:nm], "mean": mean[:nd], "bits": bits[:nd],
            "hex": [hexs[d].tobytes().decode("ascii") for d in range(nd)]}


def sense_decide(sc, msgs):
    """sense_loop's tail alone on ready-made message bodies [nmsgs][fft_size]."""
    msgs = np.ascontiguousarray(msgs, np.float32)
    nm = msgs.shape[0]
    nd = nm // (sc.avg_msgs + sc.skip_msgs)
    mean, bits, hexs = _sense_out(sc, nm, nd)
    rc = lib().orc_sense_decide(C.byref(sc), _ptr(msgs), nm, _ptr(mean), _ptr(bits), _ptr(hexs))
    if rc:
        raise ValueError("orc_sense_decide: %d" % rc)
    return {"mean": mean[:nd], "bits": bits[:nd], "hex": [hexs[d].tobytes().decode("ascii") for d in range(nd)]}
