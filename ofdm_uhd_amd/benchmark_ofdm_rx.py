#!/usr/bin/env python3
"""benchmark_ofdm_rx: IQ file -> receive_path -> packet accounting.

Mirror of the reference's benchmark_ofdm_rx.py (:35-87) with the USRP source swapped
for a file source (``--from-file``).  ``rx_callback`` is the reference's: payload[2:4]
must be 0, payload[0:2] is the packet number, packets above 19 are written to the
output file, ``n_rcvd`` / ``n_right`` are counted and printed (:50-61).
"""
import struct
import sys
from optparse import OptionParser

from . import iqio, ofdm, options as _options, receive_path


class rx_accounting(object):
    def __init__(self, packet_file=None, verbose=True):
        self.n_rcvd = 0
        self.n_right = 0
        self.packet_file = packet_file
        self.verbose = verbose

    def rx_callback(self, ok, payload):
        if len(payload) < 4:
            return  # the reference would raise struct.error on a short payload
        (preamble,) = struct.unpack('!H', payload[2:4])
        if preamble == 0:
            self.n_rcvd += 1
            (pktno,) = struct.unpack('!H', payload[0:2])
            if pktno > 19 and self.packet_file is not None:
                self.packet_file.write(payload[4:])
            if ok:
                self.n_right += 1
            if self.verbose:
                print("ok: %r \t pktno: %d \t n_rcvd: %d \t n_right: %d" % (ok, pktno, self.n_rcvd, self.n_right))


def main(argv=None):
    parser = OptionParser(option_class=_options.eng_option, conflict_handler="resolve")
    expert_grp = parser.add_option_group("Expert")
    parser.add_option("", "--snr", type="eng_float", default=30, help="set the SNR of the channel in dB [default=%default]")
    parser.add_option("", "--from-file", default="ofdm_tx.dat", help="IQ file to demodulate [default=%default]")
    parser.add_option("", "--to-file", default="rx1.txt", help="write received file contents here [default=%default]")
    parser.add_option("", "--chunk-samples", type="eng_float", default=0,
                      help="stream the capture through the demodulator in chunks of this many samples "
                           "(0 = one call on the whole file) [default=%default]")
    receive_path.receive_path.add_options(parser, expert_grp)
    ofdm.ofdm_demod.add_options(parser, expert_grp)
    (options, args) = parser.parse_args(argv)
    if len(args) != 0:
        parser.print_help(sys.stderr)
        sys.exit(1)

    packet_file = open(options.to_file, 'wb')
    acct = rx_accounting(packet_file)
    rxpath = receive_path.receive_path(acct.rx_callback, options)
    rxpath.run(iqio.file_source(options.from_file), chunk_samples=int(options.chunk_samples))
    packet_file.close()
    return acct


if __name__ == '__main__':
    try:
        main()
    except KeyboardInterrupt:
        pass
