"""Packet framing for the OFDM modem: host-side (Python 3, ``bytes``) mirror of
the reference's ``ofdm_packet_utils.py``.

Same names, arguments and error behaviour as the reference
(ofdm_packet_utils.py:29-191).  The batched product path frames packets on the
GPU (``ofdm_make_packets`` in libofdm_hip.so); these functions are the
per-packet host utilities the reference exposes alongside, and the tests check
the GPU framing against them.
"""
import math
import struct
import zlib

import numpy

from ._constants import RANDOM_MASK_HEX

# ofdm_packet_utils.py:195-453
random_mask_tuple = tuple(bytes.fromhex(RANDOM_MASK_HEX))
random_mask_vec8 = numpy.array(random_mask_tuple, numpy.uint8)


def conv_packed_binary_string_to_1_0_string(s):
    """b'\\xAF' --> '10101111' (ofdm_packet_utils.py:29-40)"""
    return "".join("{:08b}".format(x) for x in bytes(s))


def is_1_0_string(s):
    return isinstance(s, str) and all(ch in "01" for ch in s)


def conv_1_0_string_to_packed_binary_string(s):
    """'10101111' -> (b'\\xAF', False); pads with leading zeros to a multiple of 8
    and reports whether it had to (ofdm_packet_utils.py:42-72)."""
    if not is_1_0_string(s):
        raise ValueError("Input must be a string containing only 0's and 1's")
    padded = False
    rem = len(s) % 8
    if rem != 0:
        s = "0" * (8 - rem) + s
        padded = True
    return bytes(int(s[i:i + 8], 2) for i in range(0, len(s), 8)), padded


def string_to_hex_list(s):
    return [hex(x) for x in bytes(s)]


def whiten(s, o):
    """XOR with random_mask[o : o+len(s)] (ofdm_packet_utils.py:84-87)."""
    sa = numpy.frombuffer(bytes(s), numpy.uint8)
    z = sa ^ random_mask_vec8[o:len(sa) + o]
    return z.tobytes()


def dewhiten(s, o):
    return whiten(s, o)  # self inverse


def make_header(payload_len, whitener_offset=0):
    """Upper nibble is the whitener offset, lower 12 bits the length; sent twice
    (ofdm_packet_utils.py:93-97)."""
    val = ((whitener_offset & 0xf) << 12) | (payload_len & 0x0fff)
    return struct.pack('!HH', val, val)


def crc32(s):
    """digital_crc32 (digital_swig.py:3151-3169): reflected 0xEDB88320, init and
    final xor 0xFFFFFFFF -- the zlib polynomial."""
    return zlib.crc32(bytes(s)) & 0xffffffff


def gen_and_append_crc32(s):
    return bytes(s) + struct.pack(">I", crc32(s))


def check_crc32(s):
    s = bytes(s)
    if len(s) < 4:
        return (False, b'')
    msg = s[:-4]
    (expected,) = struct.unpack(">I", s[-4:])
    return (crc32(msg) == expected, msg)


def _npadding_bytes(pkt_byte_len, samples_per_symbol, bits_per_symbol):
    """Padding so the modulated packet is a multiple of 128 samples
    (ofdm_packet_utils.py:145-166)."""
    modulus = 128
    lcm = (modulus // 8) * samples_per_symbol // math.gcd(modulus // 8, samples_per_symbol)
    byte_modulus = lcm * bits_per_symbol // samples_per_symbol
    r = pkt_byte_len % byte_modulus
    if r == 0:
        return 0
    return byte_modulus - r


def make_packet(payload, samples_per_symbol, bits_per_symbol,
                pad_for_usrp=True, whitener_offset=0, whitening=True):
    """header | whiten(payload | CRC32 | 0x55 [| 0x55 pad]) (ofdm_packet_utils.py:99-143)."""
    if not (0 <= whitener_offset < 16):
        raise ValueError("whitener_offset must be between 0 and 15, inclusive (%i)" % (whitener_offset,))
    payload_with_crc = gen_and_append_crc32(payload)
    L = len(payload_with_crc)
    MAXLEN = len(random_mask_tuple)
    if L > MAXLEN:
        raise ValueError("len(payload) must be in [0, %d]" % (MAXLEN,))
    pkt_hd = make_header(L, whitener_offset)
    pkt_dt = payload_with_crc + b'\x55'
    packet_length = len(pkt_hd) + len(pkt_dt)
    if pad_for_usrp:
        pkt_dt = pkt_dt + _npadding_bytes(packet_length, samples_per_symbol, bits_per_symbol) * b'\x55'
    if whitening:
        pkt = pkt_hd + whiten(pkt_dt, whitener_offset)
    else:
        pkt = pkt_hd + pkt_dt
    return pkt


def unmake_packet(whitened_payload_with_crc, whitener_offset=0, dewhitening=1):
    """Return (ok, payload) (ofdm_packet_utils.py:169-191)."""
    if dewhitening:
        payload_with_crc = dewhiten(whitened_payload_with_crc, whitener_offset)
    else:
        payload_with_crc = bytes(whitened_payload_with_crc)
    return check_crc32(payload_with_crc)
