"""transmit_path: ofdm_mod followed by the transmit amplitude (transmit_path.py:35-84)."""
import copy

from . import ofdm


class transmit_path(object):
    def __init__(self, options, device_id=0, apply_carrier_map=False):
        """``apply_carrier_map=True`` restores the line the reference has commented out
        (transmit_path.py:67): send_pkt's carrier_map_new is then really handed to the mapper."""
        self._apply_carrier_map = bool(apply_carrier_map)
        options = copy.copy(options)    # make a copy so we can destructively modify

        self._verbose = getattr(options, "verbose", False)
        self._tx_amplitude = options.tx_amplitude                       # digital amplitude sent to the sink
        self._samples_per_symbol = getattr(options, "samples_per_symbol", 2)

        self.ofdm_tx = ofdm.ofdm_mod(options, msgq_limit=4, pad_for_usrp=False, device_id=device_id)
        self.set_tx_amplitude(self._tx_amplitude)
        self.carrier_map_old = ""
        if self._verbose:
            self._print_verbage()

    def connect(self, sink):
        self.ofdm_tx.connect(sink)
        return self

    def set_tx_amplitude(self, ampl):
        """
        Sets the transmit amplitude
        @param: ampl 0 <= ampl < 1.
        """
        self._tx_amplitude = max(0.0, min(ampl, 1))
        # the amp block (gr.multiply_const_cc) is fused into the modulator's store
        self.ofdm_tx.engine().set_tx_amplitude(self._tx_amplitude)

    def send_pkt(self, payload='', eof=False, carrier_map_new="FE7F"):
        # the reference remembers the requested map but never applies it
        # (reset_carrier_map is commented out, transmit_path.py:66-68)
        if carrier_map_new != self.carrier_map_old:
            if self._apply_carrier_map:
                self.ofdm_tx.reset_carrier_map(carrier_map_new)
            self.carrier_map_old = carrier_map_new
        return self.ofdm_tx.send_pkt(payload, eof)

    def flush(self):
        return self.ofdm_tx.flush()

    def add_options(normal, expert):
        normal.add_option("", "--tx-amplitude", type="eng_float", default=0.250, metavar="AMPL",
                          help="set transmitter digital amplitude: 0 <= AMPL < 1 [default=%default]")
        normal.add_option("-v", "--verbose", action="store_true", default=False)
        expert.add_option("-S", "--samples-per-symbol", type="int", default=2,
                          help="set samples/symbol [default=%default]")
        expert.add_option("", "--log", action="store_true", default=False,
                          help="Log all parts of flow graph to file (CAUTION: lots of data)")

    # Make a static method to call before instantiation
    add_options = staticmethod(add_options)

    def _print_verbage(self):
        print("Tx amplitude     %s" % (self._tx_amplitude))
        print("samples/symbol:  %3d" % (self._samples_per_symbol))
