"""receive_path: hands the sample stream to ofdm_demod (receive_path.py:29-58)."""
import copy

from . import ofdm


class receive_path(object):
    def __init__(self, rx_callback, options, device_id=0):
        options = copy.copy(options)    # make a copy so we can destructively modify

        self._verbose = getattr(options, "verbose", False)
        self._log = getattr(options, "log", False)
        self._rx_callback = rx_callback      # this callback is fired when there's a packet available

        self.ofdm_rx = ofdm.ofdm_demod(options, callback=self._rx_callback, device_id=device_id)

        if self._verbose:
            self._print_verbage()

    def work(self, iq):
        return self.ofdm_rx.work(iq)

    def run(self, source, chunk_samples=None):
        """Demodulate a whole source; with ``chunk_samples`` it is streamed through ofdm_demod.feed
        in pieces of that size (same packets, bounded memory)."""
        if not chunk_samples or not hasattr(source, "read_chunks"):
            return self.ofdm_rx.run(source)
        out = []
        for piece in source.read_chunks(int(chunk_samples)):
            out += self.ofdm_rx.feed(piece)
        out += self.ofdm_rx.flush()
        return out

    def feed(self, iq, flush=False):
        """Continuous operation (what the radio source does in the reference): next chunk in,
        packets that became final out; see ofdm_demod.feed."""
        return self.ofdm_rx.feed(iq, flush)

    def flush(self):
        return self.ofdm_rx.flush()

    def add_options(normal, expert):
        normal.add_option("-v", "--verbose", action="store_true", default=False)
        expert.add_option("-S", "--samples-per-symbol", type="int", default=2,
                          help="set samples/symbol [default=%default]")
        expert.add_option("", "--log", action="store_true", default=False,
                          help="Log all parts of flow graph to files (CAUTION: lots of data)")

    add_options = staticmethod(add_options)

    def _print_verbage(self):
        print("\nReceive Path:")
