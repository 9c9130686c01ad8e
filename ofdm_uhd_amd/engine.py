"""Thin Python wrapper over the C ABI of libofdm_hip.so (include/ofdm_hip.h).

``Engine`` owns one ``ofdm_handle`` (one GPU, one HIP stream).  In host mode it
takes / returns NumPy arrays; in device mode (``device_ptrs=True``) the bulk
arguments are raw device pointers (e.g. ``torch.Tensor.data_ptr()``), which is
what bench.py uses so that nothing crosses PCIe inside the timed region.

There is no CPU implementation behind this class: if the library is missing,
importing fails loudly (see _abi.load).
"""
import ctypes as C

import numpy as np

from . import _abi, config


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libofdm_hip: %s (code %d)" % (msg, code))
        self.code = code


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_payloads(payloads):
    """list of bytes -> (blob uint8, offsets uint64, lengths uint32)"""
    lens = np.array([len(p) for p in payloads], np.uint32)
    offs = np.zeros(max(len(payloads), 1), np.uint64)
    if len(payloads) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bytes(p) for p in payloads), np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)
    return np.ascontiguousarray(blob), offs[:len(payloads)] if len(payloads) else offs[:0], lens


class Engine(object):
    def __init__(self, options=None, cfg=None, pad_for_usrp=False, device_ptrs=False, device_id=0, **cfg_kw):
        self._lib = _abi.load()
        if cfg is None:
            cfg = config.make_cfg(options, pad_for_usrp=pad_for_usrp, device_ptrs=device_ptrs,
                                  device_id=device_id, **cfg_kw)
        self.cfg = cfg
        self.device_ptrs = bool(cfg.flags & _abi.OFDM_F_DEVICE_PTRS)
        self._h = C.c_void_p(None)
        rc = self._lib.ofdm_create(C.byref(cfg), C.byref(self._h))
        if rc != _abi.OFDM_OK:
            msg = self._lib.ofdm_last_error(None)
            self._h = C.c_void_p(None)
            if rc == _abi.OFDM_E_INVAL:
                raise ValueError((msg or b"").decode())
            raise EngineError(rc, (msg or b"").decode())
        self.N = cfg.fft_length
        self.CP = cfg.cp_length
        self.L = self.N + self.CP
        self.occ = cfg.occupied_tones
        self.last_stats = {}
        self._rx_sense_cfg = None

    # -- plumbing ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ofdm_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == _abi.OFDM_OK:
            return
        msg = (self._lib.ofdm_last_error(self._h) or b"").decode()
        if rc == _abi.OFDM_E_INVAL:
            raise ValueError(msg)
        raise EngineError(rc, msg)

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.ofdm_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_tx_amplitude(self, ampl):
        self._check(self._lib.ofdm_set_tx_amplitude(self._h, float(ampl)))

    def set_carrier_map(self, carriers="FE7F"):
        """reset_carrier_map of the reference's patched mapper (transmit_path.py:67): swap the data
        carrier map of BOTH directions of this engine, e.g. to hex_conv's output clipped to
        occupied_tones/4 digits (sensing_and_tramsmitting.py:470)."""
        self._check(self._lib.ofdm_set_carrier_map(self._h, (carriers or "").encode("ascii")))
        self.cfg.carrier_map = (carriers or "").encode("ascii")   # the host copy follows the handle (stream geometry)

    def set_channel(self, sigma=0.0, cfo=0.0, seed=0xC0FFEE, stream_id=0, lead=0, tail=0, enable=True):
        if not enable:
            self._check(self._lib.ofdm_set_channel(self._h, None))
            return
        ch = _abi.ofdm_chan(sigma=sigma, cfo=cfo, seed=seed, stream_id=stream_id, lead_samples=lead,
                            tail_samples=tail)
        self._check(self._lib.ofdm_set_channel(self._h, C.byref(ch)))

    def set_taps(self, *taps):
        mask = 0
        for t in taps:
            mask |= 1 << t
        self._check(self._lib.ofdm_set_taps(self._h, mask))

    def prof_enable(self, on=True):
        self._check(self._lib.ofdm_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        self._check(self._lib.ofdm_prof_reset(self._h))

    def prof(self):
        out = {}
        for k in range(_abi.K_COUNT):
            ms = C.c_double(0)
            n = C.c_uint64(0)
            self._check(self._lib.ofdm_prof_get(self._h, k, C.byref(ms), C.byref(n)))
            out[self._lib.ofdm_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    # -- framing ----------------------------------------------------------------
    def framed_len(self, payload_len):
        n = C.c_uint32(0)
        self._check(self._lib.ofdm_framed_len(self._h, int(payload_len), C.byref(n)))
        return n.value

    def make_packets(self, payloads):
        """Batched make_packet; host mode only.  Returns list of bytes."""
        assert not self.device_ptrs
        blob, offs, lens = pack_payloads(payloads)
        total = 0
        for ln in lens:
            total += self.framed_len(int(ln))
        out = np.zeros(max(total, 1), np.uint8)
        foff = np.zeros(len(payloads) + 1, np.uint64)
        self._check(self._lib.ofdm_make_packets(self._h, _ptr(blob), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                lens.ctypes.data_as(C.POINTER(C.c_uint32)), len(payloads), _ptr(out),
                                                len(out), foff.ctypes.data_as(C.POINTER(C.c_uint64))))
        return [out[int(foff[i]):int(foff[i + 1])].tobytes() for i in range(len(payloads))]

    # -- TX ----------------------------------------------------------------------
    def tx_frame_count(self, lens):
        lens = np.ascontiguousarray(lens, np.uint32)
        nsym = C.c_uint64(0)
        nsamp = C.c_uint64(0)
        self._check(self._lib.ofdm_tx_frame_count(self._h, lens.ctypes.data_as(C.POINTER(C.c_uint32)), len(lens),
                                                  C.byref(nsym), C.byref(nsamp)))
        return nsym.value, nsamp.value

    def tx(self, payloads):
        """Host mode: list of payload bytes -> complex64 IQ (incl. channel lead/tail if set)."""
        assert not self.device_ptrs
        blob, offs, lens = pack_payloads(payloads)
        _, nsamp = self.tx_frame_count(lens)
        iq = np.zeros(max(nsamp, 1), np.complex64)
        ns = C.c_uint64(0)
        st = _abi.ofdm_stats()
        self._check(self._lib.ofdm_tx(self._h, _ptr(blob), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      lens.ctypes.data_as(C.POINTER(C.c_uint32)), len(payloads), _ptr(iq), len(iq),
                                      C.byref(ns), C.byref(st)))
        self.last_stats = st.as_dict()
        return iq[:ns.value]

    def wait(self):
        """Block until everything queued on the handle's stream (tx_device(wait=False)) is done."""
        self._check(self._lib.ofdm_wait(self._h))

    def tx_device(self, payload_ptr, offs, lens, iq_ptr, iq_cap, wait=True):
        """Device mode: payload bytes and IQ are device pointers; offs/lens are NumPy host arrays.
        ``wait=False`` only queues the work (ofdm_tx_async): a following rx_device() on the same engine
        is ordered behind it on the stream, so a TX -> RX loopback needs no host round trip in between."""
        assert self.device_ptrs
        ns = C.c_uint64(0)
        st = _abi.ofdm_stats()
        fn = self._lib.ofdm_tx if wait else self._lib.ofdm_tx_async
        self._check(fn(self._h, C.c_void_p(payload_ptr), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      lens.ctypes.data_as(C.POINTER(C.c_uint32)), len(lens), C.c_void_p(iq_ptr),
                                      int(iq_cap), C.byref(ns), C.byref(st)))
        self.last_stats = st.as_dict()
        return ns.value

    def channel(self, iq, sigma=0.0, cfo=0.0, seed=0xC0FFEE, stream_id=0, index0=0):
        """Host mode: returns a new array with the synthetic channel applied."""
        assert not self.device_ptrs
        out = np.ascontiguousarray(iq, np.complex64).copy()
        ch = _abi.ofdm_chan(sigma=sigma, cfo=cfo, seed=seed, stream_id=stream_id, lead_samples=0, tail_samples=0)
        self._check(self._lib.ofdm_channel(self._h, _ptr(out), len(out), C.byref(ch), index0))
        return out

    # -- RX ----------------------------------------------------------------------
    def rx(self, iq, max_pkts=None, payload_cap=None):
        """Host mode: complex64 IQ -> list of (ok, payload) in stream order, exactly the pairs the
        reference hands to its rx callback (ofdm.py:300-305)."""
        assert not self.device_ptrs
        iq = np.ascontiguousarray(iq, np.complex64)
        if max_pkts is None:
            max_pkts = len(iq) // self.L + 16
        if payload_cap is None:
            payload_cap = max_pkts * 64 + len(iq)  # bits never exceed 8 per sample
        pay = np.zeros(max(payload_cap, 1), np.uint8)
        off = np.zeros(max_pkts + 1, np.uint64)
        ln = np.zeros(max(max_pkts, 1), np.uint32)
        ok = np.zeros(max(max_pkts, 1), np.uint8)
        npk = C.c_int(0)
        st = _abi.ofdm_stats()
        rc = self._lib.ofdm_rx(self._h, _ptr(iq) if len(iq) else None, len(iq), _ptr(pay), len(pay),
                               off.ctypes.data_as(C.POINTER(C.c_uint64)), ln.ctypes.data_as(C.POINTER(C.c_uint32)),
                               ok.ctypes.data_as(C.POINTER(C.c_uint8)), max_pkts, C.byref(npk), C.byref(st))
        self.last_stats = st.as_dict()
        self._check(rc)
        return [(bool(ok[i]), pay[int(off[i]):int(off[i]) + int(ln[i])].tobytes()) for i in range(npk.value)]

    def snr(self):
        """digital_ofdm_frame_acquisition.snr() (digital_swig.py:4231-4239): GNU Radio 3.6.0 never updates the estimate
        it initialises to 0."""
        v = C.c_float(-1.0)
        self._check(self._lib.ofdm_rx_snr(self._h, C.byref(v)))
        return float(v.value)

    def rx_submit_device(self, iq_ptr, nsamples):
        """Queue the receiver's input stage for this buffer and return at once (ofdm_rx_submit): a tx_device(...,
        wait=False) issued next is held back only until that stage has read the buffer, and runs beside the
        rx_device() call that follows with the same arguments."""
        assert self.device_ptrs
        self._check(self._lib.ofdm_rx_submit(self._h, C.c_void_p(iq_ptr), int(nsamples)))

    def rx_device(self, iq_ptr, nsamples, payload_ptr, payload_cap, max_pkts):
        """Device mode.  Returns (npkt, off, len, ok) with NumPy metadata arrays."""
        assert self.device_ptrs
        off = np.zeros(max_pkts + 1, np.uint64)
        ln = np.zeros(max(max_pkts, 1), np.uint32)
        ok = np.zeros(max(max_pkts, 1), np.uint8)
        npk = C.c_int(0)
        st = _abi.ofdm_stats()
        rc = self._lib.ofdm_rx(self._h, C.c_void_p(iq_ptr), int(nsamples), C.c_void_p(payload_ptr), int(payload_cap),
                               off.ctypes.data_as(C.POINTER(C.c_uint64)), ln.ctypes.data_as(C.POINTER(C.c_uint32)),
                               ok.ctypes.data_as(C.POINTER(C.c_uint8)), int(max_pkts), C.byref(npk), C.byref(st))
        self.last_stats = st.as_dict()
        self._check(rc)
        n = npk.value
        return n, off[:n + 1], ln[:n], ok[:n]

    # -- chunked streams --------------------------------------------------------------
    def rx_packet_pos(self):
        """Flag sample (relative to the last rx() call's IQ) of every packet it delivered."""
        n = C.c_int(0)
        self._check(self._lib.ofdm_rx_packet_pos(self._h, None, 0, C.byref(n)))
        pos = np.zeros(max(n.value, 1), np.uint64)
        if n.value:
            self._check(self._lib.ofdm_rx_packet_pos(self._h, _ptr(pos), n.value, C.byref(n)))
        return pos[:n.value]

    def rx_nco_state(self):
        """(flags uint64, phase uint64 in 2^-64 turn, step float64, swallowed uint8) of the last rx() call."""
        n = C.c_int(0)
        self._check(self._lib.ofdm_rx_nco_state(self._h, None, None, None, None, 0, C.byref(n)))
        k = n.value
        fl, phi = np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.uint64)
        st, sw = np.zeros(max(k, 1), np.float64), np.zeros(max(k, 1), np.uint8)
        if k:
            self._check(self._lib.ofdm_rx_nco_state(self._h, _ptr(fl), _ptr(phi), _ptr(st), _ptr(sw), k, C.byref(n)))
        return fl[:k], phi[:k], st[:k], sw[:k]

    def set_origin(self, first_sample_index=0):
        """Index, in its capture, of the first sample of the following rx() calls: keeps the channel
        filter's block grid where one call on the whole capture would have it."""
        self._check(self._lib.ofdm_rx_set_origin(self._h, int(first_sample_index)))

    def set_flag_history(self, flags=None, steps=None, swallowed=None, trust_after=-1, pred=(0, 0, 0.0)):
        """The settled past for the following rx() calls (flags=None switches it off): ``flags`` /
        ``steps`` / ``swallowed`` replace what a call detects up to ``trust_after``; ``pred`` = (flag,
        phase, step) of the flag before them (may lie before the call's first sample)."""
        if flags is None:
            self._check(self._lib.ofdm_rx_set_flag_history(self._h, 0, 0, None, None, None, 0, 0, 0, 0.0))
            return
        fl = np.ascontiguousarray(flags, np.int64)
        st = np.ascontiguousarray(steps, np.float64)
        sw = np.ascontiguousarray(swallowed, np.uint8)
        k = len(fl)
        self._check(self._lib.ofdm_rx_set_flag_history(self._h, 1, k, _ptr(fl) if k else None, _ptr(st) if k else None,
                                                       _ptr(sw) if k else None, int(trust_after), int(pred[0]),
                                                       int(pred[1]), float(pred[2])))

    # -- spectrum sensing ----------------------------------------------------------
    def _sense_outputs(self, sc, nm, nd):
        S = sc.fft_size
        return (np.zeros((max(nm, 1), S), np.float32), np.zeros((max(nd, 1), S), np.float64),
                np.zeros((max(nd, 1), S), np.uint8), np.zeros((max(nd, 1), S // 4), np.uint8))

    @staticmethod
    def _sense_pack(msgs, mean, bits, hexs, nm, nd):
        return {"msgs": msgs[:nm], "mean": mean[:nd], "bits": bits[:nd],
                "hex": [hexs[d].tobytes().decode("ascii") for d in range(nd)]}

    def sense_count(self, sc, nsamples):
        nm, nd = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.ofdm_sense_count(C.byref(sc), int(nsamples), C.byref(nm), C.byref(nd)))
        return nm.value, nd.value

    def sense(self, sc, iq, nsamples=None):
        """The `sensor` flowgraph + sense_loop of predictive_sense.py over one IQ stream.
        Returns {"msgs": float32[nmsgs][fft] (bin_statistics_f message bodies, FFT order),
        "mean": float64[ndec][fft], "bits": uint8[ndec][fft] (both ascending frequency),
        "hex": [str]} -- one hex carrier map per decision, as hex_conv returns it."""
        if self.device_ptrs:
            ptr, n = C.c_void_p(int(iq)), int(nsamples)
        else:
            iq = np.ascontiguousarray(iq, np.complex64)
            ptr, n = (_ptr(iq) if len(iq) else None), len(iq)
        nm, nd = self.sense_count(sc, n)
        msgs, mean, bits, hexs = self._sense_outputs(sc, nm, nd)
        onm, ond = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.ofdm_sense(self._h, C.byref(sc), ptr, n, _ptr(msgs), max(nm, 1), _ptr(mean), _ptr(bits),
                                         _ptr(hexs), max(nd, 1), C.byref(onm), C.byref(ond)))
        return self._sense_pack(msgs, mean, bits, hexs, onm.value, ond.value)

    def sense_decide(self, sc, msgs):
        """sense_loop alone over message bodies [nmsgs][fft_size] (e.g. from a real msg_queue)."""
        msgs = np.ascontiguousarray(msgs, np.float32)
        nm = msgs.shape[0]
        nd = nm // (sc.avg_msgs + sc.skip_msgs)
        _, mean, bits, hexs = self._sense_outputs(sc, 0, nd)
        ond = C.c_uint64(0)
        self._check(self._lib.ofdm_sense_decide(self._h, C.byref(sc), _ptr(msgs), nm, _ptr(mean), _ptr(bits),
                                                _ptr(hexs), max(nd, 1), C.byref(ond)))
        r = self._sense_pack(msgs, mean, bits, hexs, nm, ond.value)
        del r["msgs"]
        return r

    def set_rx_sense(self, sc):
        """Fuse the sensor into every following rx()/rx_device() call (None switches it off)."""
        self._rx_sense_cfg = sc
        self._check(self._lib.ofdm_set_rx_sense(self._h, C.byref(sc) if sc is not None else None))

    def rx_sense_result(self, nsamples, want_msgs=True, want_mean=True):
        """Outcome of the sensing run fused into the last rx()/rx_device() call.  The message
        bodies and the means are large for long streams: leave them on the device with
        want_msgs/want_mean=False when only the decisions are needed."""
        sc = self._rx_sense_cfg
        if sc is None:
            raise ValueError("set_rx_sense() has not been called")
        nm, nd = self.sense_count(sc, nsamples)
        msgs, mean, bits, hexs = self._sense_outputs(sc, nm if want_msgs else 0, nd)
        onm, ond = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.ofdm_rx_sense_result(self._h, _ptr(msgs) if want_msgs else None, max(nm, 1),
                                                   _ptr(mean) if want_mean else None, _ptr(bits), _ptr(hexs),
                                                   max(nd, 1), C.byref(onm), C.byref(ond)))
        r = self._sense_pack(msgs, mean, bits, hexs, onm.value if want_msgs else 0, ond.value)
        if not want_msgs:
            del r["msgs"]
        if not want_mean:
            del r["mean"]
        return r

    def sense_device_msgs(self):
        """(device pointer, nmsgs, fft_size) of the last run's message bodies, for an in-place
        cross-GPU max-reduce (parallel.allreduce_sensed); follow with sense_redecide()."""
        p, nm, S = C.c_void_p(None), C.c_uint64(0), C.c_uint32(0)
        self._check(self._lib.ofdm_sense_device_msgs(self._h, C.byref(p), C.byref(nm), C.byref(S)))
        return (p.value or 0), nm.value, S.value

    def sense_redecide(self, sc=None):
        sc = self._rx_sense_cfg if sc is None else sc
        self._check(self._lib.ofdm_sense_redecide(self._h, C.byref(sc)))

    # -- taps ---------------------------------------------------------------------
    _TAP_DTYPES = {
        _abi.TAP_TX_PACKETS: np.uint8, _abi.TAP_TX_FREQ: np.complex64, _abi.TAP_RX_CHAN_FILT: np.complex64,
        _abi.TAP_RX_METRIC: np.float32, _abi.TAP_RX_PEAKS: np.uint64, _abi.TAP_RX_ANGLES: np.float32,
        _abi.TAP_RX_FRAMES: np.uint64, _abi.TAP_RX_FFT: np.complex64, _abi.TAP_RX_ACQ: np.complex64,
        _abi.TAP_RX_SINK: np.complex64, _abi.TAP_RX_PACKETS: np.uint8, _abi.TAP_TX_MAPPER: np.complex64,
        _abi.TAP_TX_IFFT: np.complex64, _abi.TAP_RX_SAMPLER: np.complex64, _abi.TAP_RX_SIGMIX: np.complex64,
        _abi.TAP_RX_NCO: np.complex64, _abi.TAP_RX_PRESEL: np.float32, _abi.TAP_RX_DEMAPPED: np.uint8,
    }

    def tap(self, tap):
        nb = C.c_uint64(0)
        self._check(self._lib.ofdm_tap(self._h, tap, None, 0, C.byref(nb)))
        dt = np.dtype(self._TAP_DTYPES[tap])
        out = np.zeros(nb.value // dt.itemsize, dt)
        if nb.value:
            self._check(self._lib.ofdm_tap(self._h, tap, _ptr(out), nb.value, C.byref(nb)))
        if tap == _abi.TAP_RX_FRAMES:
            out = out.reshape(-1, 2)
        elif tap in (_abi.TAP_TX_FREQ, _abi.TAP_RX_FFT, _abi.TAP_TX_MAPPER, _abi.TAP_TX_IFFT, _abi.TAP_RX_SAMPLER):
            out = out.reshape(-1, self.N)
        elif tap in (_abi.TAP_RX_ACQ, _abi.TAP_RX_SINK):
            out = out.reshape(-1, self.occ)
        return out


def device_count():
    return _abi.load().ofdm_device_count()
