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


def test_cooperative_sensing_max_world_size_2(tmp_path):
    """Two gloo ranks sense different streams with the oracle, max-reduce the message bodies and
    take the decision on the union: a tone only rank 1 hears is flagged busy on both."""
    script = tmp_path / "worker_sense.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch
        from ofdm_uhd_amd import parallel, config
        from oracle import oracle as orc
        rank, local_rank, world = parallel.init_process_group("gloo")
        S = 256
        sc = config.make_sense_cfg(S, 1, 4, 3, 1)
        n = 5 * S * 4
        rng = np.random.default_rng(rank)
        iq = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1e-4).astype(np.complex64)
        k = 40 + 60 * rank                     # each rank hears its own tone
        iq += (0.0005 * np.exp(2j * np.pi * k * np.arange(n) / S)).astype(np.complex64)
        own = orc.sense(sc, iq)
        fused = parallel.allreduce_sensed(torch.from_numpy(own["msgs"].copy()))
        dec = orc.sense_decide(sc, fused.numpy())
        print(json.dumps({"rank": rank, "own": own["bits"][0].tolist(), "fused": dec["bits"][0].tolist(),
                          "hex": dec["hex"][0], "ge": bool((fused.numpy() >= own["msgs"]).all())}))
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
    res = [json.loads(o[0].strip().splitlines()[-1]) for o in outs]
    res.sort(key=lambda r: r["rank"])
    b0, b1 = (40 + 128) % 256, (100 + 128) % 256          # in-order positions of the two tones
    assert res[0]["own"][b0] == 0 and res[0]["own"][b1] == 1
    assert res[1]["own"][b0] == 1 and res[1]["own"][b1] == 0
    for r in res:
        assert r["ge"] and r["fused"][b0] == 0 and r["fused"][b1] == 0
        assert r["fused"] == [a & b for a, b in zip(res[0]["own"], res[1]["own"])]
    assert res[0]["hex"] == res[1]["hex"]
