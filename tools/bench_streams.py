"""Throughput with K independent streams (engines) driven concurrently on ONE GPU: python tools/bench_streams.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ofdm_uhd_amd import config, engine, options
import bench as B

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
TOTAL, size = 65536, 1026
opt = options.default_options(modulation="qpsk", tx_amplitude=0.25)
N, CP = 512, 128
L = N + CP
ncar = len(config.carrier_map(200, N))
psig = ncar / float(N) * 0.25 ** 2
sigma = float(np.sqrt(psig / 1000.0))
for K in (1, 2, 3, 4):
    P = TOTAL // K
    ctx = []
    for s in range(K):
        eng = engine.Engine(cfg=config.make_cfg(opt, device_ptrs=True))
        eng.set_channel(sigma=sigma, seed=0xC0FFEE, stream_id=s, lead=2 * N, tail=L + 2 * N)
        blob = B.make_payload_blob(P, size, s)
        offs = (np.arange(P, dtype=np.uint64) * np.uint64(size)); lens = np.full(P, size, np.uint32)
        nsym, nsamp = eng.tx_frame_count(lens)
        ctx.append(dict(eng=eng, blob=torch.from_numpy(blob.copy()).to(dev), offs=offs, lens=lens, nsym=nsym, nsamp=nsamp,
                        iq=torch.empty(nsamp * 2, dtype=torch.float32, device=dev),
                        out=torch.empty(P * size + 4096, dtype=torch.uint8, device=dev), ok=0))
    def run(c, steps):
        for _ in range(steps):
            n = c["eng"].tx_device(c["blob"].data_ptr(), c["offs"], c["lens"], c["iq"].data_ptr(), c["nsamp"])
            npk, off, ln, ok = c["eng"].rx_device(c["iq"].data_ptr(), n, c["out"].data_ptr(), c["out"].numel(), P + 1024)
            c["ok"] = int(ok.sum())
    def all_run(steps):
        th = [threading.Thread(target=run, args=(c, steps)) for c in ctx]
        for t in th: t.start()
        for t in th: t.join()
    all_run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 8
    all_run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = sum(c["nsym"] for c in ctx) * steps
    print("K=%d streams x %d packets: %.3e symbols/s  (%.2f ms per %d-packet step), crc ok %s" % (
        K, P, tot / dt, 1e3 * dt / steps, TOTAL, [c["ok"] for c in ctx]), flush=True)
    for c in ctx:
        c["eng"].close()
    del ctx
    torch.cuda.empty_cache()
