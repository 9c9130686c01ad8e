"""Gray-coded square/rectangular QAM constellations (4, 8, 16, 64, 256 points).

Same tables as the reference's ``qam.py`` (qam.py:29-73).  For a ``k``-bit symbol
value the MSB is the sign of I, the next bit the sign of Q, and the remaining
bits alternate I, Q as Gray-coded amplitude levels ``1 + 2*gray2bin(bits)``.
The table is scaled so that the largest coordinate is 1.  Values are checked
against tests/golden/reference_constants.json.
"""


def _gray_to_level(bits):
    """Gray code (MSB first) -> odd amplitude level 1, 3, 5, ..."""
    acc = 0
    value = 0
    for b in bits:
        acc ^= b
        value = (value << 1) | acc
    return 2 * value + 1


def make_constellation(m):
    k = m.bit_length() - 1
    pts = []
    peak = 1
    for i in range(m):
        bit = [(i >> (k - 1 - j)) & 1 for j in range(k)]
        re = (2 * bit[0] - 1) * _gray_to_level(bit[2::2])
        im = (2 * bit[1] - 1) * _gray_to_level(bit[3::2])
        peak = max(peak, re, im)
        pts.append((re, im))
    return [complex(re / float(peak), im / float(peak)) for re, im in pts]


constellation = {m: make_constellation(m) for m in (4, 8, 16, 64, 256)}

# symbol values are already Gray coded: identity maps (qam.py:76-113)
binary_to_gray = {m: list(range(m)) for m in (4, 8, 16, 64, 256)}
gray_to_binary = {m: list(range(m)) for m in (4, 8, 16, 64, 256)}
binary_to_ungray = {m: list(range(m)) for m in (4, 8, 16, 64)}
ungray_to_binary = {m: list(range(m)) for m in (4, 8, 16, 64)}
