"""The reference's call sequence, end to end, through the Python mirror:
benchmark_ofdm_tx -> IQ file -> benchmark_ofdm_rx (BASELINE config 1: N=512, occ=200, BPSK,
file loopback, UHD swapped for files)."""
import numpy as np
import pytest

from ofdm_uhd_amd import benchmark_ofdm_rx, benchmark_ofdm_tx, iqio, ofdm, options, receive_path, transmit_path

pytestmark = pytest.mark.gpu


def test_benchmark_file_loopback(tmp_path):
    src = tmp_path / "tx1.txt"
    data = np.random.default_rng(0).integers(0, 256, 50000, dtype=np.uint8).tobytes()
    src.write_bytes(data)
    iqf = str(tmp_path / "ofdm_tx.dat")
    out = str(tmp_path / "rx1.txt")
    npk = benchmark_ofdm_tx.main(["--from-file", str(src), "--to-file", iqf, "-M", "1.0", "-s", "1024"])
    assert npk == 20 + 49                                             # 20 garbage + ceil(50000/1022)
    iq = iqio.read_complex_binary(iqf)
    assert len(iq) % 640 == 0 and iq.dtype == np.complex64
    # the receiver needs a noise floor (0/0 in the reference's metric) and a tail that flushes the filter
    rng = np.random.default_rng(1)
    x = np.concatenate([np.zeros(1024, np.complex64), iq, np.zeros(2048, np.complex64)])
    x += ((rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x))) * 1e-3).astype(np.complex64)
    iqio.file_sink(iqf).write(x)
    acct = benchmark_ofdm_rx.main(["--from-file", iqf, "--to-file", out])
    assert acct.n_rcvd == npk and acct.n_right == npk
    assert open(out, "rb").read() == data
    # the same capture streamed in 100k-sample chunks: same file out
    out2 = str(tmp_path / "rx2.txt")
    acct2 = benchmark_ofdm_rx.main(["--from-file", iqf, "--to-file", out2, "--chunk-samples", "100k"])
    assert (acct2.n_rcvd, acct2.n_right) == (npk, npk) and open(out2, "rb").read() == data


def test_ofdm_mod_demod_objects():
    o = options.default_options(modulation="qpsk", tx_amplitude=0.25)
    got = []
    tx = transmit_path.transmit_path(o)
    sink = iqio.vector_sink()
    tx.connect(sink)
    for i in range(5):
        tx.send_pkt(b"packet %d" % i)
    with pytest.raises(ValueError):
        tx.send_pkt(b"x" * 5000)                                      # ofdm_packet_utils.py:125-126
    tx.send_pkt(eof=True)
    iq = sink.data()
    assert np.abs(iq).max() < 1.0 and 0.1 < np.sqrt(np.mean(np.abs(iq) ** 2)) < 0.2
    tx.set_tx_amplitude(0.5)                                           # doubles the samples exactly
    for i in range(5):
        tx.send_pkt(b"packet %d" % i)
    iq2 = tx.flush()
    assert np.array_equal(iq2, iq * np.float32(2.0))
    rx = receive_path.receive_path(lambda ok, p: got.append((ok, p)), o)
    rng = np.random.default_rng(2)
    x = np.concatenate([np.zeros(1024, np.complex64), iq, np.zeros(2048, np.complex64)])
    x += ((rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x))) * 1e-3).astype(np.complex64)
    rx.work(x)
    assert got == [(True, b"packet %d" % i) for i in range(5)]
    # ofdm_mod alone has unit amplitude: 4x the transmit_path output at 0.25
    m = ofdm.ofdm_mod(o, pad_for_usrp=False)
    for i in range(5):
        m.send_pkt(b"packet %d" % i)
    assert np.allclose(m.flush(), iq * 4.0, atol=1e-6)
