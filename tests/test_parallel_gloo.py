"""The N>1 path on CPU: two gloo ranks shard the streams and reduce the counters."""
import os
import socket
import subprocess
import sys
import textwrap

from ofdm_uhd_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_sharding_is_a_partition():
    for ns, w in ((8, 8), (8, 2), (7, 4), (1, 2), (16, 8)):
        parts = [parallel.streams_of_rank(ns, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(ns))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_counter_reduce_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from ofdm_uhd_amd import parallel
        rank, local_rank, world = parallel.init_process_group("gloo")
        mine = parallel.streams_of_rank(8, rank, world)
        stats = {"symbols": 22 * 100 * len(mine), "packets": 100 * len(mine), "crc_ok": 100 * len(mine) - rank,
                 "samples": 14080 * 100 * len(mine), "frames": 101 * len(mine), "peaks": 101 * len(mine)}
        parallel.barrier()
        tot = parallel.reduce_counters(stats)
        tmax = parallel.reduce_max(1.0 + rank)
        if rank == 0:
            print(json.dumps({"tot": tot, "tmax": tmax, "world": world}))
    """ % ROOT))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2 and res["tmax"] == 2.0
    assert res["tot"]["symbols"] == 22 * 100 * 8 and res["tot"]["packets"] == 800 and res["tot"]["crc_ok"] == 799
