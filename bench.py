#!/usr/bin/env python3
"""bench.py -- OFDM symbols/s, TX + loopback RX, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic input on every rank:
`packets` payloads (benchmark_ofdm_tx layout, 1026 B) -> make_packet -> map -> IFFT -> CP
(+ AWGN at 30 dB fused into the store) -> channel filter -> Schmidl-Cox sync -> CP strip ->
FFT -> equalise -> demap -> deframe/CRC, at BASELINE config 2 (N_fft=512, occ=200, CP=128,
QPSK).  Payload bytes are resident in HBM before the timed region and the recovered payloads
stay in HBM (nothing crosses PCIe inside it).  Each rank owns an independent IQ stream
(stream id = rank): weak scaling, no data-path collective; RCCL only reduces the packet
counters and the elapsed time at the end.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

# BASELINE.json configs that fit one GPU.  c2 is the headline (configs[1]); c3 / c5 use the largest legal packet
# (4 091-byte payload); c5's occ / CP are the survey's defaults (BASELINE gives only N and the modulation).
CONFIGS = {
    "c2": dict(N=512, occ=200, CP=128, mod="qpsk", size=1026, packets=65536, snr=30.0,
               name="N_fft=512, occ=200, QPSK, CP=128, synthetic AWGN channel (BASELINE configs[1])"),
    "c3": dict(N=2048, occ=1200, CP=512, mod="qam16", size=4091, packets=16384, snr=40.0,
               name="N_fft=2048, occ=1200, 16-QAM, CP=512 (BASELINE configs[2])"),
    "c5": dict(N=4096, occ=2400, CP=1024, mod="qam64", size=4091, packets=16384, snr=45.0,
               name="N_fft=4096, occ=2400, 64-QAM, CP=1024, predictive_sense FFT fused into RX (BASELINE configs[4])",
               sense=True),
}


def make_payload_blob(npkt, size, stream_id):
    """payload = !H pktno | !H 0 | data (benchmark_ofdm_tx.py:117); data from PCG64(0x0FD30000 + stream)."""
    rng = np.random.Generator(np.random.PCG64(0x0FD30000 + stream_id))
    blob = rng.integers(0, 256, size=(npkt, size), dtype=np.uint8)
    pktno = np.arange(npkt, dtype=np.uint32) & 0xFFFF
    blob[:, 0] = (pktno >> 8).astype(np.uint8)
    blob[:, 1] = (pktno & 0xFF).astype(np.uint8)
    blob[:, 2] = 0
    blob[:, 3] = 0
    return np.ascontiguousarray(blob.reshape(-1))


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box
    shows all host cores in the mask but grants a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(round(float(parts[0]) / float(parts[1])))))
            else:
                q = float(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(round(q / float(f.read())))))
            break
        except (IOError, OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 16))


def cpu_baseline(cfg, sigma, lead, tail, size, npkt):
    """The oracle (a C port of the reference flow graph, one thread per stream) on a bounded sample of the
    same workload, timed on this host's cores: first one stream on one core, then one independent stream per
    core on all cores of the box's share.  A reported baseline, not the optimisation target.
    The single-stream sample's IQ is then pushed through the GPU receiver as the checker: same packets, same
    verdicts."""
    import threading
    from oracle import oracle as orc
    from ofdm_uhd_amd import engine
    orc.lib()

    def one_stream(stream_id, n, out):
        blob = make_payload_blob(n, size, stream_id)
        pay = [blob[i * size:(i + 1) * size].tobytes() for i in range(n)]
        iq = orc.tx(cfg, pay, lead=lead, tail=tail)
        orc.channel(iq, sigma=sigma, seed=0xC0FFEE, stream_id=stream_id)
        r = orc.rx(cfg, iq)
        nsym = (len(iq) - lead - tail) // (cfg.fft_length + cfg.cp_length)
        out[stream_id] = (nsym, sum(1 for o, _ in r.packets if o), iq if stream_id == 0 else None,
                          r.packets if stream_id == 0 else None)

    # (a) one stream, one core
    res = {}
    t0 = time.perf_counter()
    one_stream(0, npkt, res)
    dt1 = time.perf_counter() - t0
    nsym1, ok1, iq, pkts = res[0]
    eng = engine.Engine(cfg=cfg)
    same = eng.rx(iq) == pkts
    eng.close()
    del iq
    # (b) one stream per core (ctypes releases the GIL inside the oracle), half of the sample each
    cores = host_cores()
    per = max(256, npkt // 2)
    resn = {}
    th = [threading.Thread(target=one_stream, args=(i + 1, per, resn)) for i in range(cores)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dtn = time.perf_counter() - t0
    nsymn = sum(v[0] for v in resn.values())
    okn = sum(v[1] for v in resn.values())
    return {"value": nsymn / dtn, "unit": "OFDM symbols/s", "cores": cores, "kind": "port",
            "sample": "%d independent streams x %d packets (%d symbols) of the same workload through "
                      "oracle/ofdm_oracle.c, TX+AWGN+RX, one thread per stream, %.1f s; CRC pass %d/%d"
                      % (cores, per, nsymn, dtn, okn, cores * per),
            "single_core": {"value": nsym1 / dt1, "cores": 1,
                            "sample": "%d packets (%d symbols), %.1f s; CRC pass %d/%d" % (npkt, nsym1, dt1, ok1, npkt)},
            "gpu_rx_matches_on_sample": bool(same)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--packets", type=int, default=None, help="packets per stream per step (default: per config)")
    ap.add_argument("--size", type=int, default=None, help="payload bytes (default: per config)")
    ap.add_argument("--snr", type=float, default=None)
    ap.add_argument("--cpu-packets", type=int, default=4096, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--sense", default="auto", choices=("auto", "on", "off"),
                    help="fuse the predictive_sense.py spectrum sensor into every RX call (auto: on for c5)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run TX and RX of every step strictly one after the other (default: the next step's TX is queued "
                         "behind the input stage of this step's RX, on the handle's transmit stream)")
    ap.add_argument("--sync", default="pn", choices=("pn", "fixed"),
                    help="receiver front end: 'pn' = the reference's Schmidl-Cox chain (default, the headline number); "
                         "'fixed' = its known-timing test mode (ofdm_receiver.py~:108-119): no filter, no metric -- "
                         "times the rest of the receiver on its own")
    ap.add_argument("--iq-buffers", type=int, default=None, choices=(1, 2),
                    help="2: steps alternate two IQ buffers (the engine orders a transmit batch only behind reads of the "
                         "buffer it writes): step i+1's modulator may run beside step i's channel filter, and beside a "
                         "receiver that reads its input to the end of the call (fused sensing, SYNC 'fixed': pipelining "
                         "needs it there).  Default: 2 with --sync fixed, else 1 (at c2 two buffers measure the same, at c5 "
                         "pipelining measures slower)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="BASELINE.json config to run (default c2 = configs[1], the one the metric is quoted on)")
    args = ap.parse_args()

    import torch

    from ofdm_uhd_amd import _abi, config, engine, options, parallel

    # OFDM_BENCH_REHEARSE=1: rehearse the multi-rank path on a ONE-GPU box -- gloo instead of RCCL and every
    # rank on cuda:0 (RCCL refuses two ranks on one device).  Never set by the driver; numbers are meaningless.
    rehearse = os.environ.get("OFDM_BENCH_REHEARSE") == "1"
    rank, local_rank, world = parallel.init_process_group("gloo" if rehearse else None)
    if rehearse:
        local_rank = 0
    if world != max(args.gpus, 1) and rank == 0:
        sys.stderr.write("warning: --gpus %d but WORLD_SIZE %d\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine is the only implementation)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfgd = CONFIGS[args.config]
    N, occ, CP, mod = cfgd["N"], cfgd["occ"], cfgd["CP"], cfgd["mod"]
    if args.packets is None:
        args.packets = cfgd["packets"]
    if args.size is None:
        args.size = cfgd["size"]
    if args.snr is None:
        args.snr = cfgd["snr"]
    opt = options.default_options(modulation=mod, fft_length=N, occupied_tones=occ, cp_length=CP, tx_amplitude=0.25)
    L = N + CP
    P, size = args.packets, args.size
    if args.sync == "fixed":
        # ofdm_sync_fixed flags the last sample of every nsymbols-th symbol counted from the first sample: the stream
        # starts on a preamble (no lead-in) and every packet has the same length
        probe = engine.Engine(cfg=config.make_cfg(opt, device_ptrs=True, device_id=local_rank))
        nsym1, _ = probe.tx_frame_count(np.full(1, size, np.uint32))
        probe.close()
        opt.sync, opt.sync_nsymbols, opt.sync_freq_offset = "fixed", int(nsym1), 0.0
    cfg = config.make_cfg(opt, device_ptrs=True, device_id=local_rank)
    eng = engine.Engine(cfg=cfg)
    stream_id = rank
    ncar = len(config.carrier_map(occ, N))
    # mean in-packet power: ncar carriers of mean constellation power through IFFT/sqrt(N), amplitude 0.25
    cpow = float(np.mean(np.abs(np.array(config.rotated_constellation(mod))) ** 2))
    psig = ncar / float(N) * 0.25 ** 2 * cpow
    sigma = float(np.sqrt(psig / 10 ** (args.snr / 10.0)))
    lead, tail = (0, 2 * N) if args.sync == "fixed" else (2 * N, L + 2 * N)
    eng.set_channel(sigma=sigma, cfo=0.0, seed=0xC0FFEE, stream_id=stream_id, lead=lead, tail=tail)

    blob = make_payload_blob(P, size, stream_id)
    offs = (np.arange(P, dtype=np.uint64) * np.uint64(size))
    lens = np.full(P, size, np.uint32)
    nsym, nsamp = eng.tx_frame_count(lens)
    d_blob = torch.from_numpy(blob).to(dev)                       # resident before the timed region
    # (sense_on is what the configuration says unless --sense overrides it)
    reads_to_end = (cfgd.get("sense", False) if args.sense == "auto" else args.sense == "on") or args.sync == "fixed"
    if args.iq_buffers is None:
        # measured: SYNC 'fixed' at c2 5.05 ms pipelined over two buffers against 5.7 in sequence; c5 (fused sensing)
        # 9.7-9.8 against 9.45 -- its occupancy-starved kernels lose more to the modulator beside them than the overlap
        # gains -- so c5 stays in sequence unless --iq-buffers 2 is given
        args.iq_buffers = 2 if (args.sync == "fixed" and not args.no_pipeline) else 1
    d_iqs = [torch.empty(nsamp * 2, dtype=torch.float32, device=dev) for _ in range(args.iq_buffers)]
    d_out = torch.empty(P * size + 4096, dtype=torch.uint8, device=dev)
    max_pkts = P + 1024

    # BASELINE config 5: the spectrum sensor of predictive_sense.py rides on the receiver's IQ buffer.  FFT size =
    # the OFDM FFT size; tune / dwell as sensor.__init__ derives them (1 ms / 10 ms at 6.25 MS/s, in FFT frames);
    # 10 messages averaged + 1 consumed per decision, threshold 1e-4.
    sense_on = cfgd.get("sense", False) if args.sense == "auto" else args.sense == "on"
    sc = None
    sense_tot = {"messages": 0, "decisions": 0}
    last_hex = [None]
    if sense_on:
        sc = config.make_sense_cfg(N, max(0, int(round(1e-3 * 6.25e6 / N))), max(1, int(round(10e-3 * 6.25e6 / N))),
                                   10, 1, 0.00010)
        eng.set_rx_sense(sc)

    # One step = TX of a batch + RX of that batch.  The handle's transmit side has a stream of its own: the NEXT
    # step's TX is queued as soon as this step's receiver has read the IQ buffer (its input stage, rx_submit) and
    # runs beside the rest of this step's RX, filling the receiver's host round trips.  Every step's TX and RX lie
    # inside the timed region.  With fused sensing (c5) or SYNC 'fixed' the receiver reads the IQ buffer to the end of
    # the call (the engine refuses a transmit batch into it before that): the next batch then goes into a SECOND IQ
    # buffer (steps alternate two), or there is no overlap.
    pipelined = not args.no_pipeline and (not reads_to_end or args.iq_buffers >= 2)

    def tx(i):
        n = eng.tx_device(d_blob.data_ptr(), offs, lens, d_iqs[i % len(d_iqs)].data_ptr(), nsamp, wait=False)
        return n, dict(eng.last_stats)

    def run_steps(nsteps, tot, pipelined=pipelined):
        npk = off = ln = ok = None
        if nsteps <= 0:
            return npk, off, ln, ok
        n, tx_stats = tx(0)
        for i in range(nsteps):
            d_iq = d_iqs[i % len(d_iqs)]
            if pipelined:
                eng.rx_submit_device(d_iq.data_ptr(), n)
                nxt = tx(i + 1) if i + 1 < nsteps else None
            npk, off, ln, ok = eng.rx_device(d_iq.data_ptr(), n, d_out.data_ptr(), d_out.numel(), max_pkts)
            rx_stats = dict(eng.last_stats)
            if sense_on:
                if world > 1:
                    # cooperative sensing: every GPU decides on the max over all antennas (one small all_reduce)
                    parallel.allreduce_sensed(eng.sense_device_msgs(), device=dev)
                    torch.cuda.synchronize()
                    eng.sense_redecide()
                res = eng.rx_sense_result(n, want_msgs=False, want_mean=False)
                nm, nd = eng.sense_count(sc, n)
                sense_tot["messages"] += nm
                sense_tot["decisions"] += len(res["hex"])
                last_hex[0] = res["hex"][-1] if res["hex"] else None
            if tot is not None:
                tot["symbols"] += tx_stats["symbols"]
                tot["samples"] += rx_stats["samples"]
                tot["packets"] += npk
                tot["crc_ok"] += int(ok.sum())
                tot["frames"] += rx_stats["frames"]
                tot["peaks"] += rx_stats["peaks"]
            if not pipelined and i + 1 < nsteps:
                nxt = tx(i + 1)
            if i + 1 < nsteps:
                n, tx_stats = nxt
        eng.wait()
        return npk, off, ln, ok

    run_steps(args.warmup, None)
    sense_tot["messages"] = sense_tot["decisions"] = 0
    eng.prof_enable(True)
    eng.prof_reset()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = {"symbols": 0, "packets": 0, "crc_ok": 0, "samples": 0, "frames": 0, "peaks": 0}
    npk, off, ln, ok = run_steps(args.steps, tot)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.prof()
    # Per-kernel durations.  While steps overlap, a HIP-event span around a kernel also counts the time it shared the
    # GPU with the other stream's kernels; the durations the roofline is priced on are therefore measured on a few
    # extra steps run strictly in sequence right after the timed region (same buffers, same work; not part of `value`).
    prof_overlapped = None
    if pipelined:
        prof_overlapped = prof
        eng.prof_reset()
        nseq = max(2, min(4, args.steps))
        run_steps(nseq, None, pipelined=False)
        prof = eng.prof()
        prof_steps = nseq
    else:
        prof_steps = max(args.steps, 1)
    eng.prof_enable(False)

    # correctness of the last step: every delivered payload whose CRC passed is bit-exact what was sent
    # (the reference's own timing jitter loses about one packet in 10^4 at 30 dB; the oracle shows the same)
    delivered_ok = bool(npk == P and bool((ln == size).all()))
    if delivered_ok:
        got = d_out[:P * size].view(P, size)
        sent = d_blob.view(P, size)
        row_eq = (got == sent).all(dim=1).cpu().numpy()
        delivered_ok = bool(np.all(row_eq[ok.astype(bool)]))
    elapsed = parallel.reduce_max(elapsed, device=dev)
    g = parallel.reduce_counters(tot, device=dev)
    all_exact = parallel.reduce_counters({"packets": int(delivered_ok)}, device=dev)["packets"] == world

    result_line = None
    if rank == 0:
        sym_per_s = g["symbols"] / elapsed
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        # dominant kernel by HIP-event time (events recorded on the engine's own stream)
        kern = max(prof.items(), key=lambda kv: kv[1][0])
        kname, (kms, klaunch) = kern
        # algorithmic bytes of ONE launch of that kernel (DESIGN.md "Roofline accounting"):
        #   k_chan_filter / k_sync / k_rx_demod : the compulsory read of the received stream, (N+CP)*8 B per symbol
        #   k_tx_mod            : the compulsory write of the stream + the packet bits in
        nbits = int(np.ceil(np.log2(cfg.arity)))
        bits_b = ncar * nbits / 8.0                                    # payload bits of one symbol, in bytes
        per_symbol = {"k_sync": L * 8.0, "k_chan_filter": L * 8.0, "k_rx_demod": N * 8.0 + bits_b,
                      "k_tx_mod": L * 8.0 + bits_b}.get(kname, L * 8.0)
        launch_bytes = per_symbol * nsym
        avg_s = (kms / max(klaunch, 1)) * 1e-3
        achieved = launch_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        path_bytes = 2 * L * 8.0 + 2 * bits_b                        # 10 339 B per symbol at C2
        # HBM traffic of that kernel from the committed PMC passes of this same command (cannot be collected from
        # inside the process): read + write bytes per symbol x symbols of one launch; null for other configs
        # (tools/make_pmc_traffic.py stamps the file with a hash of the kernel sources: a stale file is not quoted)
        traffic, ktraffic, traffic_src = None, None, None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from make_pmc_traffic import sources_sha
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config)) as f:
                pmc = json.load(f)
            bps = pmc["bytes_per_symbol"].get(kname)
            if pmc.get("sources_sha") != sources_sha():
                traffic_src = "stale: %s was collected on other kernel sources" % pmc.get("source")
            else:
                # the whole path: every kernel's read + write bytes per symbol x the symbols of one step
                traffic = sum(v["read"] + v["write"] for v in pmc["bytes_per_symbol"].values()) * nsym
                if bps:
                    ktraffic = (bps["read"] + bps["write"]) * nsym
                traffic_src = pmc["source"]
        except (ImportError, IOError, OSError, ValueError, KeyError):
            pass
        path_achieved = sym_per_s / world * path_bytes / 1e9
        out = {
            "metric": "OFDM symbols/sec (TX+loopback RX) @ N_fft=512; packet CRC pass rate",
            "value": sym_per_s,
            "unit": "OFDM symbols/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfgd["name"] + ("" if args.sync == "pn" else " -- SYNC='fixed' test mode (no filter / metric)"),
                       "packets_per_stream_per_step": P, "payload_bytes": size, "symbols_per_packet": nsym // P,
                       "snr_db": args.snr, "streams": world, "iq_buffers": args.iq_buffers, "parallelism": "independent streams, 1 per GPU",
                       "pipelining": (("TX of step i+1 queued on the handle's transmit stream behind the input stage of "
                                       "step i's RX" if not reads_to_end else
                                       "TX of step i+1 into the other of two IQ buffers, beside step i's RX (which reads its "
                                       "buffer to the end of the call)") if pipelined else "none (steps in sequence)")},
            "crc_pass_rate": g["crc_ok"] / float(max(world * P * args.steps, 1)),
            "crc_ok_payloads_bit_exact": all_exact,
            # (the synthetic channel's Gaussian is a 16-bit-radius Box-Muller, truncated at 4.85 sigma: pass rates at
            #  high SNR are marginally optimistic against an untruncated Gaussian)
            "channel": "AWGN %.1f dB fused into the TX store; Philox-2x32-7, Box-Muller with 16-bit radius (|n| <= 4.85 sigma)" % args.snr,
            # `achieved` / `frac`: the PATH figure the north star's 40 % is defined on -- SURVEY 8(d)'s algorithmic bytes per
            # symbol (TX writes the stream once, RX reads it once, payload bits in and out) x symbols per second per GPU.
            # kernel_*: the dominant kernel on its own compulsory bytes over its own HIP-event duration.
            "roofline": {"bound": "hbm", "achieved": path_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": path_achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_symbol": path_bytes, "algorithmic_bytes_per_step": path_bytes * nsym,
                         "kernel": kname, "kernel_achieved": achieved, "kernel_frac": achieved / HBM_PEAK_GBPS,
                         "kernel_avg_ms": kms / max(klaunch, 1), "kernel_algorithmic_bytes_per_launch": launch_bytes,
                         "kernel_traffic": ktraffic,
                         "kernel_timing": ("HIP events on the kernel's stream, %d steps run in sequence right after the timed "
                                           "region (inside it TX and RX of neighbouring steps overlap)" % prof_steps) if pipelined
                                          else "HIP events on the kernel's stream over the timed region",
                         "path_achieved": path_achieved, "path_frac": path_achieved / HBM_PEAK_GBPS},
            "kernels_ms_per_step": {k: v[0] / prof_steps for k, v in prof.items()},
        }
        if prof_overlapped is not None:
            out["kernels_ms_per_step_overlapped"] = {k: v[0] / max(args.steps, 1) for k, v in prof_overlapped.items()}
        if sense_on:
            sms, sl = prof.get("k_sense", (0.0, 0))
            # k_sense reads every IQ sample of the accrued vectors once: 8 B/sample
            used = (sense_tot["messages"] // max(args.steps, 1)) * sc.dwell_delay * sc.fft_size * 8.0
            out["sensing"] = {"fft_size": sc.fft_size, "tune_delay": sc.tune_delay, "dwell_delay": sc.dwell_delay,
                              "messages_per_step": sense_tot["messages"] // max(args.steps, 1),
                              "decisions_per_step": sense_tot["decisions"] // max(args.steps, 1),
                              "last_carrier_map_head": (last_hex[0] or "")[:64],
                              "k_sense_avg_ms": sms / max(sl, 1),
                              "k_sense_GBps": used / (sms / max(sl, 1) * 1e-3) / 1e9 if sms > 0 else None,
                              "overlapped_with": "the peak-detector pass, after k_sync (second HIP stream)",
                              "fusion": "max over ranks (all_reduce)" if world > 1 else "single antenna"}
        if args.cpu_packets > 0 and world == 1:      # the CPU leg runs at N=1 only (rank 0 would hold the others up)
            cfg_host = config.make_cfg(opt)
            out["cpu_baseline"] = cpu_baseline(cfg_host, sigma, lead, tail, size, args.cpu_packets)
        else:
            out["cpu_baseline"] = None
        result_line = json.dumps(out)
    eng.close()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass
    if result_line is not None:
        # last thing on stdout (RCCL / gloo print their own banners there too)
        sys.stdout.flush()
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
