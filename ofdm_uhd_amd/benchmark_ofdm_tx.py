#!/usr/bin/env python3
"""benchmark_ofdm_tx: packet source -> transmit_path -> IQ file.

Mirror of the reference's benchmark_ofdm_tx.py (:38-126) with the USRP sink swapped
for a file sink (``--to-file``): the first 20 packets carry "This is Garbage data",
then the input file is sent in chunks of ``size - 2`` bytes; every payload is
``!H pktno | !H 0 | data`` (benchmark_ofdm_tx.py:111-117).
"""
import os
import struct
import sys
from optparse import OptionParser

from . import iqio, ofdm, options as _options, transmit_path


def build_payloads(options, data_source=None):
    """The reference's packet loop (benchmark_ofdm_tx.py:97-124) as a generator."""
    nbytes = int(10e6 * options.megabytes)   # sic: 1 "megabyte" = 10e6 bytes (:97)
    n = 0
    pktno = 0
    pkt_size = int(options.size)
    preamble = 0
    while n < nbytes:
        if pktno < 20:
            data = b"This is Garbage data"
        else:
            data = data_source.read(pkt_size - 2) if data_source is not None else b''
            if data == b'':
                break
        payload = struct.pack('!H', pktno & 0xffff) + struct.pack('!H', preamble & 0xffff) + data
        yield payload
        n += len(payload)
        pktno += 1


def main(argv=None):
    parser = OptionParser(option_class=_options.eng_option, conflict_handler="resolve")
    expert_grp = parser.add_option_group("Expert")
    parser.add_option("-s", "--size", type="eng_float", default=1024, help="set packet size [default=%default]")
    parser.add_option("-M", "--megabytes", type="eng_float", default=1.0,
                      help="set megabytes to transmit [default=%default]")
    parser.add_option("", "--discontinuous", action="store_true", default=False, help="enable discontinuous mode")
    parser.add_option("", "--from-file", default=None, help="use file for packet contents")
    parser.add_option("", "--to-file", default="ofdm_tx.dat", help="write the modulated IQ here [default=%default]")
    transmit_path.transmit_path.add_options(parser, expert_grp)
    ofdm.ofdm_mod.add_options(parser, expert_grp)
    (options, args) = parser.parse_args(argv)
    if len(args) != 0:
        parser.print_help()
        sys.exit(1)

    src = open(options.from_file, 'rb') if options.from_file is not None else None
    if src is not None:
        print(os.path.getsize(options.from_file))

    txpath = transmit_path.transmit_path(options)
    sink = iqio.file_sink(options.to_file)
    txpath.connect(sink)
    npk = 0
    for payload in build_payloads(options, src):
        txpath.send_pkt(payload)
        sys.stderr.write('.')
        npk += 1
    txpath.send_pkt(eof=True)
    sink.close()
    sys.stderr.write("\n%d packets, %d symbols -> %s\n" % (npk, txpath.ofdm_tx.symbols_sent, options.to_file))
    return npk


if __name__ == '__main__':
    try:
        main()
    except KeyboardInterrupt:
        pass
