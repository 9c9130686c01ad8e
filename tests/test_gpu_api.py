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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_path_world_size_1(tmp_path):
    """The RCCL side of the multi-GPU path on the one GPU a test box has: a fresh child process
    brings up the "nccl" (= RCCL) process group with WORLD_SIZE=1 before anything touches the GPU, then runs the
    collectives the bench runs -- reduce_counters, reduce_max -- and the cooperative-sensing exchange on the engine's
    own device buffer (zero-copy all_reduce(MAX) + ofdm_sense_redecide): with one antenna the fused decisions must be
    the single-antenna ones."""
    import json
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_ws1.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        from ofdm_uhd_amd import parallel
        rank, local_rank, world = parallel.init_process_group("nccl")       # before any GPU call
        import numpy as np, torch
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        dev = torch.device("cuda", local_rank)
        from helpers import make_cfg, make_payloads
        from ofdm_uhd_amd import config, engine
        tot = parallel.reduce_counters({"symbols": 22, "packets": 3, "crc_ok": 2, "samples": 14080, "frames": 4, "peaks": 5}, device=dev)
        tmax = parallel.reduce_max(1.25, device=dev)
        parallel.barrier()
        cfg = make_cfg("qpsk")
        eng = engine.Engine(cfg=cfg)
        eng.set_channel(sigma=0.003, lead=1024, tail=1664)
        x = eng.tx(make_payloads(24, 600, seed=5))
        sc = config.make_sense_cfg(256, 1, 6, 3, 1, threshold=0.05)
        eng.set_rx_sense(sc)
        pk = eng.rx(x)
        single = eng.rx_sense_result(len(x))
        fused = parallel.allreduce_sensed(eng.sense_device_msgs(), device=dev)   # in place on the engine's buffer
        torch.cuda.synchronize()
        eng.sense_redecide()
        after = eng.rx_sense_result(len(x))
        print(json.dumps({"tot": tot, "tmax": tmax, "npk": len(pk), "nok": sum(ok for ok, _ in pk),
                          "hex_same": single["hex"] == after["hex"], "ndec": len(after["hex"]),
                          "msgs_same": bool(np.array_equal(single["msgs"], after["msgs"])),
                          "fused_same": bool(np.array_equal(fused.cpu().numpy(), single["msgs"]))}))
        eng.close()
        dist.destroy_process_group()
    """ % (root, root)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["tot"] == {"symbols": 22, "samples": 14080, "packets": 3, "crc_ok": 2, "frames": 4, "peaks": 5}
    assert res["tmax"] == 1.25 and res["npk"] == 24 and res["nok"] >= 23
    assert res["ndec"] >= 1 and res["hex_same"] and res["msgs_same"] and res["fused_same"]


def test_bench_under_torchrun_one_rank():
    """bench.py --gpus 1 launched the way the driver launches N > 1 (torch.distributed.run, one rank, RCCL process group
    up): the same JSON line as the plain launch -- same metric, same work, same recovery."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--packets", "4096", "--cpu-packets", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    a = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args,
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    ja, jb = json.loads(a.stdout.strip().splitlines()[-1]), json.loads(b.stdout.strip().splitlines()[-1])
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "config", "crc_pass_rate", "crc_ok_payloads_bit_exact"):
        assert ja[k] == jb[k], k
    assert jb["n_gpus"] == 1 and jb["crc_ok_payloads_bit_exact"]
    assert 0.5 < jb["value"] / ja["value"] < 2.0          # (4096-packet steps are launch-bound: loose on purpose)


def test_frame_acquisition_snr_accessor():
    """digital_ofdm_frame_acquisition.snr() (digital_swig.py:4231-4239): GNU Radio 3.6.0 initialises the estimate to 0
    and never updates it; the ABI carries the accessor with that behaviour."""
    from ofdm_uhd_amd import engine
    from helpers import make_cfg
    e = engine.Engine(cfg=make_cfg("qpsk"))
    assert e.snr() == 0.0
    e.close()


@pytest.mark.parametrize("extra", [["--config", "c3", "--packets", "512"],
                                   ["--config", "c5", "--packets", "512"],
                                   ["--config", "c5", "--packets", "512", "--iq-buffers", "2"],
                                   ["--sync", "fixed", "--packets", "2048"],
                                   ["--no-pipeline", "--packets", "2048"]])
def test_bench_modes_contract(extra):
    """Every bench configuration / mode (small batches): one JSON line with the contract's keys, the roofline and
    cpu_baseline objects, every CRC-ok payload bit-exact."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["bench.py", "--steps", "2", "--warmup", "1", "--cpu-packets", "64"] + extra
    p = subprocess.run([sys.executable] + args, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["scaling"] == "weak" and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_rx_matches_on_sample"]
    assert j["crc_ok_payloads_bit_exact"] and j["crc_pass_rate"] > 0.9
