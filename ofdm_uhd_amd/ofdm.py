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
import sys

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
            self._engine.set_taps(engine._abi.TAP_TX_FREQ)
        self._log = bool(getattr(options, "log", False))

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
        # same limit and exception as make_packet (ofdm_packet_utils.py:123-126)
        if len(payload) + 4 > len(ofdm_packet_utils.random_mask_tuple):
            raise ValueError("len(payload) must be in [0, %d]" % (len(ofdm_packet_utils.random_mask_tuple),))
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
        iq = self._engine.tx(self._pending)
        self.symbols_sent += self._engine.last_stats.get("symbols", 0)
        self.packets_sent += len(self._pending)
        if self._log:
            self._write_logs(iq)
        self._pending = []
        if self._sink is not None:
            self._sink.write(iq)
        return iq

    def _write_logs(self, iq):
        # the reference's --log probe points (ofdm.py:123-131)
        freq = self._engine.tap(engine._abi.TAP_TX_FREQ)
        iqio.file_sink("ofdm_preambles.dat", append=True).write(freq.reshape(-1))
        iqio.file_sink("ofdm_cp_adder_c.dat", append=True).write(iq)

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
            self._engine.set_taps(engine._abi.TAP_RX_FFT, engine._abi.TAP_RX_ACQ, engine._abi.TAP_RX_SINK)
        self.n_packets = 0
        self.n_ok = 0
        if getattr(options, "verbose", False):
            self._print_verbage()

    def engine(self):
        return self._engine

    def work(self, iq):
        """Demodulate one contiguous IQ stream; fires the callback per packet and returns the
        list of (ok, payload)."""
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

    def reset_carrier_map(self, carrier_map_new):
        """The frame sink's side of reset_carrier_map: streams demodulated from now on are
        de-mapped with the new data-carrier set."""
        self._engine.set_carrier_map(carrier_map_new)

    def last_stats(self):
        return dict(self._engine.last_stats)

    def _write_logs(self):
        A = engine._abi
        e = self._engine
        iqio.file_sink("ofdm_receiver-chan_filt_c.dat").write(e.tap(A.TAP_RX_CHAN_FILT))
        iqio.file_sink("ofdm_receiver-fft_out_c.dat").write(e.tap(A.TAP_RX_FFT).reshape(-1))
        iqio.file_sink("ofdm_receiver-frame_acq_c.dat").write(e.tap(A.TAP_RX_ACQ).reshape(-1))
        iqio.file_sink("ofdm_frame_sink_c.dat").write(e.tap(A.TAP_RX_SINK).reshape(-1))

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
