"""Shared builders for the parity tests."""
import struct

import numpy as np

from ofdm_uhd_amd import config, options


def make_cfg(mod="qpsk", N=512, occ=200, CP=128, **kw):
    opt = options.default_options(modulation=mod, fft_length=N, occupied_tones=occ, cp_length=CP)
    return config.make_cfg(opt, **kw)


def make_payloads(npkt, plen, seed=1, variant="random"):
    """benchmark_ofdm_tx payload layout: !H pktno | !H 0 | data (benchmark_ofdm_tx.py:117)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(npkt):
        n = plen if np.isscalar(plen) else int(plen[i])
        head = struct.pack('!HH', i & 0xffff, 0)[:n]
        if variant == "random":
            body = rng.integers(0, 256, max(n - 4, 0), dtype=np.uint8).tobytes()
        else:  # reference style: chr(pktno & 0xff) * n (sensing_and_tramsmitting.py:419)
            body = bytes([i & 0xff]) * max(n - 4, 0)
        out.append((head + body)[:n])
    return out


def loopback_stream(orc, cfg, payloads, snr_db=30.0, cfo_bins=0.0, lead=None, tail=None, seed=0xC0FFEE, stream_id=0):
    """Oracle TX + channel: the ONE input array both receivers are fed."""
    N, CP = cfg.fft_length, cfg.cp_length
    lead = 2 * N if lead is None else lead
    tail = (N + CP) + 2 * N if tail is None else tail
    iq = orc.tx(cfg, payloads, lead=lead, tail=tail)
    core = iq[lead:len(iq) - tail] if tail else iq[lead:]
    psig = float(np.mean(np.abs(core) ** 2)) if len(core) else 1.0
    sigma = float(np.sqrt(psig / (10 ** (snr_db / 10.0))))
    orc.channel(iq, sigma=sigma, cfo=cfo_bins * 2 * np.pi / N, seed=seed, stream_id=stream_id)
    return iq
