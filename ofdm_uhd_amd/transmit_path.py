"""Host-side counterpart of the reference's ``transmit_path`` (transmit_path.py:35-84): an
``ofdm_mod`` plus the output gain.  Same constructor argument, methods and command-line flags;
the gain stage (gr.multiply_const_cc ``amp``) is folded into the modulator kernel's store."""
import copy

from . import ofdm

# flag, group, keyword arguments of the reference's option table (transmit_path.py:72-79)
_FLAGS = (
    (("", "--tx-amplitude"), "normal", dict(type="eng_float", default=0.250, metavar="AMPL",
                                            help="digital output amplitude, 0 <= AMPL < 1 [default=%default]")),
    (("-v", "--verbose"), "normal", dict(action="store_true", default=False)),
    (("-S", "--samples-per-symbol"), "expert", dict(type="int", default=2, help="samples per symbol [default=%default]")),
    (("", "--log"), "expert", dict(action="store_true", default=False,
                                   help="dump every probe point of the flow graph to files (large)")),
)


class transmit_path(object):
    def __init__(self, options, device_id=0, apply_carrier_map=False):
        """``apply_carrier_map=True`` re-enables what transmit_path.py:67 has commented out: the map
        given to send_pkt really reaches the mapper (and must reach the receiver's frame sink too)."""
        opts = copy.copy(options)
        self._apply_carrier_map = bool(apply_carrier_map)
        self._verbose = bool(getattr(opts, "verbose", False))
        self._samples_per_symbol = getattr(opts, "samples_per_symbol", 2)
        self.carrier_map_old = ""
        self.ofdm_tx = ofdm.ofdm_mod(opts, msgq_limit=4, pad_for_usrp=False, device_id=device_id)
        self.set_tx_amplitude(opts.tx_amplitude)
        if self._verbose:
            self._print_verbage()

    # -- wiring ---------------------------------------------------------------------
    def connect(self, sink):
        """Attach the IQ sink (iqio.file_sink / vector_sink ...) that stands in for the radio."""
        self.ofdm_tx.connect(sink)
        return self

    # -- controls -------------------------------------------------------------------
    def set_tx_amplitude(self, ampl):
        """Output gain, clamped to [0, 1] as transmit_path.py:56-62 does."""
        self._tx_amplitude = min(max(ampl, 0.0), 1)
        self.ofdm_tx.engine().set_tx_amplitude(self._tx_amplitude)

    def send_pkt(self, payload='', eof=False, carrier_map_new="FE7F"):
        """Queue one packet (or end the burst with eof=True).  A changed carrier map is only recorded,
        exactly like the reference, unless the path was built with apply_carrier_map=True."""
        changed = carrier_map_new != self.carrier_map_old
        if changed and self._apply_carrier_map:
            self.ofdm_tx.reset_carrier_map(carrier_map_new)
        if changed:
            self.carrier_map_old = carrier_map_new
        return self.ofdm_tx.send_pkt(payload, eof)

    def flush(self):
        return self.ofdm_tx.flush()

    # -- command line ---------------------------------------------------------------
    @staticmethod
    def add_options(normal, expert):
        for names, group, kw in _FLAGS:
            (normal if group == "normal" else expert).add_option(*names, **kw)

    def _print_verbage(self):
        print("Tx amplitude     %s" % (self._tx_amplitude,))
        print("samples/symbol:  %3d" % (self._samples_per_symbol,))
