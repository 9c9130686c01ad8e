"""Gray-coded M-PSK constellations for M in {2, 4, 8}.

Same tables as the reference's ``psk.py`` (psk.py:27-60): point ``i`` sits at
angle ``step[i] * 2*pi/M``; the step table below is the reference's bit
formula evaluated once.  Values are checked against
tests/golden/reference_constants.json to the last float64 bit.
"""
import cmath
import math

# signed position on the circle, in units of 2*pi/M, for symbol value i
_GRAY_STEPS = {
    2: (0, 1),
    4: (0, 1, 3, 2),
    8: (0, 1, 3, 2, -1, -2, -4, -3),
}


def make_gray_constellation(m):
    unit = 2 * math.pi / m
    out = []
    for step in _GRAY_STEPS[m]:
        theta = unit * abs(step)
        if step < 0:
            theta = -theta
        out.append(complex(math.cos(theta), math.sin(theta)))
    return out


def make_constellation(m):
    """Points that simply increment around the unit circle (psk.py:49-50)."""
    return [cmath.exp(i * 2 * math.pi / m * 1j) for i in range(m)]


constellation = {m: make_constellation(m) for m in (2, 4, 8)}
gray_constellation = {m: make_gray_constellation(m) for m in (2, 4, 8)}

binary_to_gray = {2: list(range(2)), 4: [0, 1, 3, 2], 8: [0, 1, 3, 2, 7, 6, 4, 5]}
