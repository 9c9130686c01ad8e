"""ofdm_mod / ofdm_demod: packets in, baseband IQ out -- and back.

Python-3 mirror of the reference's ``ofdm.py`` (ofdm.py:38-305) with the same
constructor arguments, option names, ``send_pkt`` / callback surface and error
behaviour.  Where the reference wires GNU Radio blocks into a flow graph, this
module hands batches to the HIP engine (libofdm_hip.so): ``send_pkt`` queues a
payload, ``send_pkt(eof=True)`` (or ``flush()``) modulates the queued batch in one
GPU call and writes the samples to the connected sink; ``ofdm_demod.work(iq)``
demodulates one contiguous IQ stream and fires ``callback(ok, payload)`` once per
recovered packet, in stream order, on the caller's thread.
"""
import math
import sys

import numpy as np

from . import config, engine, iqio, ofdm_packet_utils  # noqa: F401  (ofdm_packet_utils re-exported like digital.ofdm_packet_utils)
from .config import known_symbols_4512_3  # noqa: F401  (ofdm.py:310-325)


class ofdm_mod(object):
    """
    Modulates an OFDM stream. Based on the options fft_length, occupied_tones, and
    cp_length, this block creates OFDM symbols using a specified modulation option.

    Send packets by calling send_pkt
    """

    def __init__(self, options, msgq_limit=2, pad_for_usrp=True, device_id=0):
        """
        @param options: pass modulation options from higher layers (fft length, occupied tones, etc.)
        @param msgq_limit: maximum number of messages in message queue (kept for API
               compatibility: the reference blocks send_pkt at this depth, ofdm.py:148;
               here packets are batched until flush)
        @param pad_for_usrp: If true, packets are padded such that they end up a multiple of 128 samples
        """
        self._pad_for_usrp = pad_for_usrp
        self._msgq_limit = msgq_limit
        self._modulation = options.modulation
        self._fft_length = options.fft_length
        self._occupied_tones = options.occupied_tones
        self._cp_length = options.cp_length

        # ofdm.py:71-87: preamble = first occupied_tones known symbols, odd bins zeroed
        self._ksfreq = config.make_ksfreq(self._fft_length, self._occupied_tones)
        self._padded_preambles = [config.padded_preamble(self._fft_length, self._occupied_tones)]
        self._rotated_const = config.rotated_constellation(self._modulation)  # ofdm.py:91-101

        # the modulator alone has unit gain after its 1/sqrt(N) (ofdm.py:114); transmit_path sets the amplitude
        cfg_opts = _copy_options(options, tx_amplitude=1.0)
        self._engine = engine.Engine(cfg_opts, pad_for_usrp=pad_for_usrp, device_id=device_id)
        self._pending = []
        self._sink = None
        self.symbols_sent = 0
        self.packets_sent = 0

        if getattr(options, "verbose", False):
            self._print_verbage()
        if getattr(options, "log", False):
            self._engine.set_taps(engine._abi.TAP_TX_FREQ, engine._abi.TAP_TX_MAPPER, engine._abi.TAP_TX_IFFT)
        self._log = bool(getattr(options, "log", False))
        if self._log:
            # gr.file_sink opens its file when the flow graph is built (truncating it) and appends for the life of
            # the graph (ofdm.py:123-131)
            for name in ("ofdm_mapper_c.dat", "ofdm_preambles.dat", "ofdm_ifft_c.dat", "ofdm_cp_adder_c.dat"):
                open(name, "wb").close()

    # -- wiring -------------------------------------------------------------------
    def connect(self, sink):
        """Attach the object that receives the modulated samples (``write(iq)``)."""
        self._sink = sink
        return self

    def engine(self):
        return self._engine

    # -- packets ------------------------------------------------------------------
    def send_pkt(self, payload='', eof=False):
        """
        Send the payload.

        @param payload: data to send
        @type payload: bytes (str is encoded latin-1)
        """
        if eof:
            self.flush()  # gr.message(1): no more packets (ofdm.py:142)
            return
        if isinstance(payload, str):
            payload = payload.encode("latin-1")
        payload = bytes(payload)
        # same limit and exception as make_packet (ofdm_packet_utils.py:123-126) -- and, like it, raised HERE for
        # the offending packet only: the padded, whitened body must fit the mask too (whiten(), :84-87)
        if len(payload) + 4 > len(ofdm_packet_utils.random_mask_tuple):
            raise ValueError("len(payload) must be in [0, %d]" % (len(ofdm_packet_utils.random_mask_tuple),))
        self._engine.framed_len(len(payload))        # ValueError when the whitening mask is exhausted
        self._pending.append(payload)

    def reset_carrier_map(self, carrier_map_new):
        """digital_ofdm_mapper_bcv.reset_carrier_map of the reference's patched GNU Radio
        (the call transmit_path.py:67 has commented out): packets queued so far go out on the old
        map, later ones on the new one."""
        self.flush()
        self._engine.set_carrier_map(carrier_map_new)

    def flush(self):
        """Modulate everything queued so far; returns the samples (also written to the sink)."""
        if not self._pending:
            return None
        pending, self._pending = self._pending, []   # a failing batch never poisons the queue
        iq = self._engine.tx(pending)
        self.symbols_sent += self._engine.last_stats.get("symbols", 0)
        self.packets_sent += len(pending)
        if self._log:
            self._write_logs(iq)
        if self._sink is not None:
            self._sink.write(iq)
        return iq

    def _write_logs(self, iq):
        # the reference's --log probe points (ofdm.py:123-131)
        A = engine._abi
        iqio.file_sink("ofdm_mapper_c.dat", append=True).write(self._engine.tap(A.TAP_TX_MAPPER).reshape(-1))
        iqio.file_sink("ofdm_preambles.dat", append=True).write(self._engine.tap(A.TAP_TX_FREQ).reshape(-1))
        ifft = self._engine.tap(A.TAP_TX_IFFT)
        iqio.file_sink("ofdm_ifft_c.dat", append=True).write(ifft.reshape(-1))
        # ofdm_cp_adder_c.dat: cp_adder's output, BEFORE the 1/sqrt(N) scale block (ofdm.py:113-114,130)
        cp = self._cp_length
        iqio.file_sink("ofdm_cp_adder_c.dat", append=True).write(np.concatenate([ifft[:, ifft.shape[1] - cp:], ifft], axis=1).reshape(-1))

    def add_options(normal, expert):
        """
        Adds OFDM-specific options to the Options Parser
        """
        normal.add_option("-m", "--modulation", type="string", default="bpsk",
                          help="set modulation type (bpsk, qpsk, 8psk, qam{16,64}) [default=%default]")
        expert.add_option("", "--fft-length", type="intx", default=512,
                          help="set the number of FFT bins [default=%default]")
        expert.add_option("", "--occupied-tones", type="intx", default=200,
                          help="set the number of occupied FFT bins [default=%default]")
        expert.add_option("", "--cp-length", type="intx", default=128,
                          help="set the number of bits in the cyclic prefix [default=%default]")
    # Make a static method to call before instantiation
    add_options = staticmethod(add_options)

    def _print_verbage(self):
        """
        Prints information about the OFDM modulator
        """
        print("\nOFDM Modulator:")
        print("Modulation Type: %s" % (self._modulation))
        print("FFT length:      %3d" % (self._fft_length))
        print("Occupied Tones:  %3d" % (self._occupied_tones))
        print("CP length:       %3d" % (self._cp_length))


class ofdm_demod(object):
    """
    Demodulates a received OFDM stream. Based on the options fft_length, occupied_tones, and
    cp_length, this block performs synchronization, FFT, and demodulation of incoming OFDM
    symbols and passes packets up the a higher layer.

    The input is complex baseband.  When packets are demodulated, they are passed to the
    app via the callback.
    """

    def __init__(self, options, callback=None, device_id=0):
        """
        @param options: pass modulation options from higher layers (fft length, occupied tones, etc.)
        @param callback:  function of two args: ok, payload
        @type callback: ok: bool; payload: bytes
        """
        self._modulation = options.modulation
        self._fft_length = options.fft_length
        self._occupied_tones = options.occupied_tones
        self._cp_length = options.cp_length
        self._snr = getattr(options, "snr", 30)
        self._callback = callback

        self._ksfreq = config.make_ksfreq(self._fft_length, self._occupied_tones)  # ofdm.py:210-215
        self._rotated_const = config.rotated_constellation(self._modulation)       # ofdm.py:225-236
        self._engine = engine.Engine(options, device_id=device_id)
        self._log = bool(getattr(options, "log", False))
        if self._log:
            self._engine.set_taps(engine._abi.TAP_RX_FFT, engine._abi.TAP_RX_ACQ, engine._abi.TAP_RX_SINK,
                                  engine._abi.TAP_RX_SAMPLER, engine._abi.TAP_RX_SIGMIX, engine._abi.TAP_RX_NCO)
            # file sinks: opened (truncated) with the graph, appended to for its life (ofdm_receiver.py~:144-152,
            # ofdm.py:253-254)
            for name in self._LOG_FILES.values():
                open(name, "wb").close()
        self._log_samples = 0        # samples of the capture the per-sample probe files already hold
        self.n_packets = 0
        self.n_ok = 0
        self._streaming = False      # feed() has data or history pending
        self.reset_stream()
        if getattr(options, "verbose", False):
            self._print_verbage()

    def engine(self):
        return self._engine

    def work(self, iq):
        """Demodulate one contiguous IQ stream; fires the callback per packet and returns the
        list of (ok, payload)."""
        if self._streaming:
            self.reset_stream()  # a one-shot call ends any chunked stream (and drops its carried history)
        pkts = self._engine.rx(iq)
        if self._log:
            self._write_logs()
        for ok, payload in pkts:
            self.n_packets += 1
            if ok:
                self.n_ok += 1
            if self._callback:
                self._callback(ok, payload)  # _queue_watcher_thread.run (ofdm.py:300-305)
        return pkts

    def run(self, source):
        return self.work(source.read_all())

    # -- continuous operation: the flow graph never stops, captures arrive in chunks ------------------
    #
    # The engine demodulates one contiguous array per call and starts every call like the flow graph
    # starts (zero filter / correlator history, detector average 0, NCO phase 0).  feed() stitches
    # chunks so that the packets equal those of ONE call on the whole capture:
    #   * each call sees [carried tail | new chunk]; the tail starts on the sync kernel's tile grid and
    #     reaches back far enough (the detector average looks back 34 tiles; the first tiles of a call
    #     lack correlator history; a packet begun before the horizon can swallow later frames) for
    #     everything after the previous horizon to be detected exactly as in the uncut stream;
    #   * packets whose preamble flag lies beyond `horizon` = end of data minus one maximum-length
    #     packet are held back (their symbols may continue in the next chunk) and come out of the next
    #     call; packets at or before the previous horizon were delivered already and are skipped;
    #   * the NCO continues from the last final flag (phase and step carried over).
    def _stream_geometry(self):
        cfg = self._engine.cfg
        N, CP = cfg.fft_length, cfg.cp_length
        L = N + CP
        T = 2048                                         # SYNC_TILE of csrc/rx_sync.h
        nbits = max(1, int(math.ceil(math.log(cfg.arity, 2))))
        ncar = len(config.carrier_map(cfg.occupied_tones, cfg.occupied_tones, cfg.carrier_map.decode("ascii") or "FE7F"))
        sym_max = int(math.ceil(8.0 * (4 + 4095 + 17) / (ncar * nbits))) + 1
        span = (sym_max + 3) * L + 2 * T + int(cfg.ntaps)
        lookback = (34 + 3 + (L + T - 1) // T) * T
        return T, span, lookback

    def reset_stream(self):
        self._s_tail = np.zeros(0, np.complex64)   # samples carried into the next call
        self._s_abs = 0                            # absolute index of _s_tail[0]
        self._s_final = -1                         # every flag <= this absolute index has been dealt with
        # settled flags still of interest: (abs flag, phase in 2^-64 turn, step, swallowed); before any flag
        # the NCO idles at phase 0
        self._s_hist = [(0, 0, 0.0, 0)]
        self._engine.set_flag_history(None)
        self._engine.set_origin(0)
        self._streaming = False
        self._log_samples = 0

    def feed(self, iq, flush=False):
        """Demodulate the next chunk of a continuous capture; returns the packets that became final.
        ``flush=True`` (or flush()) ends the stream: everything still held back is delivered."""
        if self._engine.cfg.sync_mode != engine._abi.SYNC_PN:
            raise ValueError("feed() needs SYNC 'pn': ofdm_sync_fixed's flags are positions in the whole capture")
        self._streaming = True
        T, span, lookback = self._stream_geometry()
        iq = np.ascontiguousarray(iq, np.complex64)
        buf = np.concatenate([self._s_tail, iq]) if len(self._s_tail) else iq
        base = self._s_abs
        total = base + len(buf)
        horizon = total if flush else total - span           # flags <= horizon are final after this call
        prev_final = self._s_final
        ran = False
        out = []
        if len(buf) and horizon > self._s_final:
            eng = self._engine
            # the settled past: flags inside this buffer keep their known steps (whatever the call re-detects
            # in its unsettled overlap is dropped); the last one before the buffer is the NCO's predecessor
            inside = [f for f in self._s_hist if f[0] >= max(base, 1)]
            before = [f for f in self._s_hist if f[0] < max(base, 1)]
            pred = before[-1] if before else (0, 0, 0.0, 0)
            eng.set_flag_history([f[0] - base for f in inside], [f[2] for f in inside], [f[3] for f in inside],
                                 trust_after=self._s_final - base, pred=(pred[0] - base, pred[1], pred[2]))
            eng.set_origin(base)
            pkts = eng.rx(buf)
            ran = True
            pos = eng.rx_packet_pos().astype(np.int64) + base
            for (ok, payload), p in zip(pkts, pos):
                if self._s_final < p <= horizon:
                    out.append((ok, payload))
            fl, phi, st, sw = eng.rx_nco_state()
            fl = fl.astype(np.int64) + base
            for j in np.flatnonzero((fl > self._s_final) & (fl <= horizon)):
                self._s_hist.append((int(fl[j]), int(phi[j]), float(st[j]), int(sw[j])))
            self._s_final = max(self._s_final, horizon)
        if self._log and ran:
            self._write_logs(base=base, prev_final=prev_final, horizon=horizon, end=total if flush else None)
        if flush:
            self.reset_stream()
        else:
            # carry: one maximum packet (a packet that began before the horizon may swallow frames after
            # it) plus the detector's look-back before the horizon, on the tile grid of the absolute stream
            start = max(base, ((horizon - lookback - span) // T) * T) if horizon > 0 else base
            self._s_tail = buf[start - base:].copy()
            self._s_abs = start
            # history: the flags of the carried part and the last one before it
            keep = [f for f in self._s_hist if f[0] >= max(start, 1)]
            older = [f for f in self._s_hist if f[0] < max(start, 1)]
            self._s_hist = older[-1:] + keep
        for ok, payload in out:
            self.n_packets += 1
            if ok:
                self.n_ok += 1
            if self._callback:
                self._callback(ok, payload)
        return out

    def flush(self):
        return self.feed(np.zeros(0, np.complex64), flush=True)

    def reset_carrier_map(self, carrier_map_new):
        """The frame sink's side of reset_carrier_map: streams demodulated from now on are
        de-mapped with the new data-carrier set."""
        self._engine.set_carrier_map(carrier_map_new)

    def last_stats(self):
        return dict(self._engine.last_stats)

    _LOG_FILES = {"chan_filt": "ofdm_receiver-chan_filt_c.dat", "fft": "ofdm_receiver-fft_out_c.dat",
                  "acq": "ofdm_receiver-frame_acq_c.dat", "sampler": "ofdm_receiver-sampler_c.dat",
                  "sigmix": "ofdm_receiver-sigmix_c.dat", "nco": "ofdm_receiver-nco_c.dat", "sink": "ofdm_frame_sink_c.dat"}

    def _write_logs(self, base=None, prev_final=None, horizon=None, end=None):
        """The reference's --log probe files (ofdm_receiver.py~:144-152, ofdm.py:253-254): appended to for the life of
        the receiver, like its gr.file_sink blocks.  A one-shot work() appends the whole call.  In a chunked stream
        (feed) every call re-processes a carried tail: only what became FINAL in this call is appended -- per-sample
        probes from the last sample written up to the horizon (the NCO behind it still waits for its flags), symbol
        rows of the frames whose flag lies in (previous horizon, horizon] -- so that the files of a chunked run equal
        those of one call on the whole capture."""
        A = engine._abi
        e = self._engine
        F = self._LOG_FILES

        def app(key, arr):
            iqio.file_sink(F[key], append=True).write(np.ascontiguousarray(arr).reshape(-1))

        y, sm, nco = e.tap(A.TAP_RX_CHAN_FILT), e.tap(A.TAP_RX_SIGMIX), e.tap(A.TAP_RX_NCO)
        fft, acq, samp, sink = e.tap(A.TAP_RX_FFT), e.tap(A.TAP_RX_ACQ), e.tap(A.TAP_RX_SAMPLER), e.tap(A.TAP_RX_SINK)
        if base is None:                      # one-shot
            if not len(sm):                   # (no flag at all: the NCO idles at phase 0, sigmix = chan_filt)
                sm, nco = y, np.ones(len(y), np.complex64)
            for key, arr in (("chan_filt", y), ("sigmix", sm), ("nco", nco), ("fft", fft), ("acq", acq),
                             ("sampler", samp), ("sink", sink)):
                app(key, arr)
            return
        # per-sample probes: absolute samples [log_samples, stop)
        stop = end if end is not None else max(horizon, self._log_samples)
        lo, hi = self._log_samples - base, stop - base
        if hi > lo >= 0:
            app("chan_filt", y[lo:hi])
            # (a call that raised no flag at all computes no NCO: phase 0 throughout, sigmix = chan_filt)
            app("sigmix", sm[lo:hi] if len(sm) else y[lo:hi])
            app("nco", nco[lo:hi] if len(nco) else np.ones(hi - lo, np.complex64))
            self._log_samples = stop
        # symbol rows: frame f of this call holds K[f] + 1 consecutive rows
        fr = e.tap(A.TAP_RX_FRAMES)
        if len(fr):
            rows = np.concatenate([[0], np.cumsum(fr[:, 1].astype(np.int64) + 1)])
            dem = e.tap(A.TAP_RX_DEMAPPED).astype(bool)
            sink_row = np.cumsum(dem) - 1         # row of RX_SINK a demapped symbol went to
            for f in range(len(fr)):
                p = int(fr[f, 0]) + base
                if prev_final < p <= horizon:
                    r0, r1 = int(rows[f]), int(rows[f + 1])
                    app("fft", fft[r0:r1])
                    app("acq", acq[r0:r1])
                    app("sampler", samp[r0:r1])
                    sel = sink_row[r0:r1][dem[r0:r1]]
                    if len(sel):
                        app("sink", sink[sel])

    def add_options(normal, expert):
        """
        Adds OFDM-specific options to the Options Parser
        """
        normal.add_option("-m", "--modulation", type="string", default="bpsk",
                          help="set modulation type (bpsk or qpsk) [default=%default]")
        expert.add_option("", "--fft-length", type="intx", default=512,
                          help="set the number of FFT bins [default=%default]")
        expert.add_option("", "--occupied-tones", type="intx", default=200,
                          help="set the number of occupied FFT bins [default=%default]")
        expert.add_option("", "--cp-length", type="intx", default=128,
                          help="set the number of bits in the cyclic prefix [default=%default]")
    # Make a static method to call before instantiation
    add_options = staticmethod(add_options)

    def _print_verbage(self):
        """
        Prints information about the OFDM demodulator
        """
        print("\nOFDM Demodulator:")
        print("Modulation Type: %s" % (self._modulation))
        print("FFT length:      %3d" % (self._fft_length))
        print("Occupied Tones:  %3d" % (self._occupied_tones))
        print("CP length:       %3d" % (self._cp_length))


def _copy_options(options, **overrides):
    import copy
    o = copy.copy(options)
    for k, v in overrides.items():
        setattr(o, k, v)
    return o
