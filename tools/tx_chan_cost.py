"""k_tx_mod time with and without the fused channel (noise generator share): python tools/tx_chan_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ofdm_uhd_amd import config, engine, options
dev = torch.device("cuda", 0)
P, size = 65536, 1026
eng = engine.Engine(cfg=config.make_cfg(options.default_options(modulation="qpsk", tx_amplitude=0.25), device_ptrs=True))
blob = np.random.default_rng(0).integers(0, 256, P * size, dtype=np.uint8)
offs = np.arange(P, dtype=np.uint64) * np.uint64(size)
lens = np.full(P, size, np.uint32)
d_blob = torch.from_numpy(blob).to(dev)
for chan in (False, True, False, True):
    eng.set_channel(sigma=0.003, lead=1024, tail=1664, enable=chan)
    nsym, nsamp = eng.tx_frame_count(lens)
    d_iq = torch.empty(nsamp * 2, dtype=torch.float32, device=dev)
    eng.prof_enable(True)
    for it in range(4):
        if it == 1:
            eng.prof_reset()
        eng.tx_device(d_blob.data_ptr(), offs, lens, d_iq.data_ptr(), nsamp)
    ms, n = eng.prof()["k_tx_mod"]
    print("channel fused=%s  k_tx_mod %.3f ms  (%d symbols, %.0f GB/s written)" % (chan, ms / n, nsym, nsym * 5120 / (ms / n * 1e-3) / 1e9))
