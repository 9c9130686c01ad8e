"""IQ sample sources and sinks that stand in for the reference's UHD radio I/O
(usrp_transmit_path.py:66-72, usrp_receive_path.py:67-73).

File format = ``gr.file_sink(gr.sizeof_gr_complex, ...)`` / ``gr.file_source``:
raw little-endian interleaved float32 I,Q (ofdm.py:124-131;
utils/read_complex_binary.m:40-45), so captures move freely between this engine
and a real GNU Radio flow graph.
"""
import numpy as np


class vector_sink(object):
    """Collects everything written to it (gr.vector_sink_c)."""

    def __init__(self):
        self._chunks = []

    def write(self, iq):
        self._chunks.append(np.ascontiguousarray(iq, np.complex64))

    def data(self):
        if not self._chunks:
            return np.zeros(0, np.complex64)
        return np.concatenate(self._chunks)

    def close(self):
        pass


class file_sink(object):
    """gr.file_sink(gr.sizeof_gr_complex, filename)"""

    def __init__(self, filename, append=False):
        self._f = open(filename, "ab" if append else "wb")

    def write(self, iq):
        np.ascontiguousarray(iq, np.complex64).astype("<c8", copy=False).tofile(self._f)

    def close(self):
        if self._f:
            self._f.close()
            self._f = None


class null_sink(object):
    def write(self, iq):
        pass

    def close(self):
        pass


def read_complex_binary(filename, count=-1, offset_samples=0):
    """utils/read_complex_binary.m: interleaved float32 -> complex64."""
    return np.fromfile(filename, dtype="<c8", count=count, offset=8 * offset_samples).astype(np.complex64, copy=False)


class file_source(object):
    """gr.file_source(gr.sizeof_gr_complex, filename, repeat) (predictive_sense.py:92)."""

    def __init__(self, filename, repeat=False):
        self.filename = filename
        self.repeat = repeat

    def read_all(self):
        return read_complex_binary(self.filename)

    def read_chunks(self, chunk_samples):
        """The file in pieces of chunk_samples (the last one shorter), without loading it whole."""
        off = 0
        while True:
            a = read_complex_binary(self.filename, count=chunk_samples, offset_samples=off)
            if len(a) == 0:
                return
            yield a
            off += len(a)
            if len(a) < chunk_samples:
                return


class vector_source(object):
    def __init__(self, iq):
        self._iq = np.ascontiguousarray(iq, np.complex64)

    def read_all(self):
        return self._iq

    def read_chunks(self, chunk_samples):
        for a in range(0, len(self._iq), chunk_samples):
            yield self._iq[a:a + chunk_samples]
