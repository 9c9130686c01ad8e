"""Independent float64 NumPy model of the OFDM hot path (test helper).

A second, separately written restatement of SURVEY.md Appendix A used to pin
the C oracle (oracle/ofdm_oracle.c): vectorised NumPy / np.fft in float64 where
the oracle loops in float32.  It also states the CLOSED FORMS the HIP engine
uses for the sequential GNU Radio blocks (sampler automaton, NCO phase), so the
tests check "closed form == automaton" on the CPU before the GPU relies on it.
"""
import math

import numpy as np

from ofdm_uhd_amd import config, ofdm_packet_utils


def cfg_arrays(cfg):
    N, occ = cfg.fft_length, cfg.occupied_tones
    const = np.array([complex(cfg.constellation[i].re, cfg.constellation[i].im) for i in range(cfg.arity)])
    ks = np.array([complex(cfg.known_symbol[i].re, cfg.known_symbol[i].im) for i in range(occ)])
    taps = np.array([cfg.taps[i] for i in range(cfg.ntaps)], np.float64)
    return N, occ, cfg.cp_length, const, ks, taps


def nbits_of(cfg):
    return int(math.ceil(math.log2(cfg.arity)))


def pad_symbol(seed, pkt, slot, arity):
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (pkt + 1) + 0xBF58476D1CE4E5B9 * (slot + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return (z >> 32) % arity


def tx_freq(cfg, payloads):
    """Mapper + insert_preamble output: list of N-vectors (complex128)."""
    N, occ, CP, const, ks, _ = cfg_arrays(cfg)
    nb = nbits_of(cfg)
    cmap = np.array(config.carrier_map(occ, N))
    zl = config.zeros_on_left(N, occ)
    pre = np.zeros(N, complex)
    pre[zl:zl + occ] = ks
    out = []
    for p, payload in enumerate(payloads):
        pkt = ofdm_packet_utils.make_packet(payload, 1, 1, bool(cfg.flags & 2), cfg.whitener_offset, True)
        bits = np.unpackbits(np.frombuffer(pkt, np.uint8), bitorder="little")
        nchunks = len(bits) // nb
        vals = (bits[:nchunks * nb].reshape(-1, nb) * (1 << np.arange(nb))).sum(axis=1)
        nsym = -(-len(bits) // (len(cmap) * nb))
        out.append(pre.copy())
        for s in range(nsym):
            v = np.zeros(N, complex)
            for i, c in enumerate(cmap):
                slot = s * len(cmap) + i
                b = vals[slot] if slot < nchunks else pad_symbol(cfg.pad_seed, p, slot, cfg.arity)
                v[c] = const[b]
            out.append(v)
    return np.array(out)


def tx_time(cfg, freq):
    """fft_vcc(inverse, shift) -> cyclic prefixer -> 1/sqrt(N) -> amplitude."""
    N, CP = cfg.fft_length, cfg.cp_length
    x = np.fft.ifft(np.fft.ifftshift(freq, axes=1), axis=1) * N  # unnormalised inverse DFT
    x = np.concatenate([x[:, N - CP:], x], axis=1)
    return (x * (1.0 / math.sqrt(N)) * float(cfg.tx_amplitude)).reshape(-1)


def chan_filter(cfg, x):
    taps = cfg_arrays(cfg)[5]
    return np.convolve(x.astype(np.complex128), taps)[:len(x)]


def sync_metric(cfg, y):
    N, CP = cfg.fft_length, cfg.cp_length
    D = N // 2
    yd = np.concatenate([np.zeros(D, complex), y[:-D]]) if len(y) > D else np.zeros(len(y), complex)
    c = y * np.conj(yd)
    e = np.abs(y) ** 2

    def movsum(a, w):
        cs = np.concatenate([[0], np.cumsum(a)])
        idx = np.arange(len(a))
        lo = np.maximum(idx + 1 - w, 0)
        return cs[idx + 1] - cs[lo]

    P = movsum(c, D)
    R = movsum(e, D)
    with np.errstate(divide="ignore", invalid="ignore"):
        M = np.where(R * R > 0, np.abs(P) ** 2 / (R * R), 0.0)
    M = np.minimum(M, 1024.0)
    Mbar = movsum(M, CP) / CP
    return Mbar - 1.0, P


def peak_detect(u, rise=0.2, fall=0.2, alpha=0.001):
    """gr_peak_detector_fb, one buffer.  avg is a plain IIR of u (every sample
    updates it exactly once), so thresholds are known up front; the state machine
    then only has work to do where u > thr or a run is open."""
    u = np.asarray(u, np.float64)
    n = len(u)
    avg = np.empty(n + 1)
    avg[0] = 0.0
    a = 0.0
    for i in range(n):  # sequential IIR (float64 here)
        a = alpha * u[i] + (1 - alpha) * a
        avg[i + 1] = a
    thr_rise = avg[:-1] * rise
    thr_fall = avg[:-1] * fall
    peaks = []
    state = 0
    pv, pi = -np.inf, 0
    i = 0
    cand = np.flatnonzero(u > np.minimum(thr_rise, thr_fall))
    ci = 0
    while i < n:
        if state == 0:
            # jump to the next sample that can start a run
            while ci < len(cand) and cand[ci] < i:
                ci += 1
            if ci == len(cand):
                break
            i = cand[ci]
            if u[i] > thr_rise[i]:
                state = 1
            else:
                i += 1
        else:
            if u[i] > pv:
                pv, pi = u[i], i
                i += 1
            elif u[i] > thr_fall[i]:
                i += 1
            else:
                peaks.append(pi)
                state = 0
                pv = -np.inf
    return np.array(peaks, np.int64)


def sampler_frames(peaks, nsamples, N, CP, timeout=1000):
    """Closed form of digital_ofdm_sampler: every flag p >= N that the automaton
    reaches starts a frame at [p-N+1, p]; data symbol k (k >= 1) ends at p + k*L and
    is emitted iff no flag lies in [p+1, p+1+k*L], k <= timeout+1 and the call had its
    L+N+1 samples.  Returns [(p, K)]."""
    L = N + CP
    frames = []
    prev = None   # (p, K) of the previous frame
    base_ns = 0   # base of the NO_SIG scan grid (steps of L+1)
    for j, p in enumerate(peaks):
        p = int(p)
        nxt = int(peaks[j + 1]) if j + 1 < len(peaks) else None
        # ---- is the flag seen, and in which call? -------------------------------
        if prev is not None and prev[2] == "frame":
            # previous frame still in FRAME state when p is scanned: found in the call whose
            # base is b = prev_p - N + 1 + (k-1)*L with p in [b+N, b+L+N]
            pp, pK = prev[0], prev[1]
            k = pK + 1                       # the call that would have emitted data symbol K+1
            b = pp - N + 1 + (k - 1) * L
        else:
            if p < base_ns + N:
                continue                     # never scanned (before the first window)
            m = (p - N - base_ns) // (L + 1)
            b = base_ns + m * (L + 1)
        if not (b + L + N < nsamples):
            break                            # the call needs L+N+1 samples: stream exhausted
        # ---- data symbols of this frame -------------------------------------------
        K = 0
        state = "frame"
        while True:
            k = K + 1
            bk = p - N + 1 + (k - 1) * L
            if not (bk + L + N < nsamples):
                state = "end"
                break
            if nxt is not None and nxt <= p + 1 + k * L:
                break                        # re-trigger found in this call
            K += 1
            if K == timeout + 1:
                state = "nosig"
                base_ns = p - N + 1 + K * L
                break
        frames.append((p, K))
        prev = (p, K, state)
        if state == "end":
            break
    return frames


def nco_phase(peaks, angles, N):
    """Phi[j] = phase of the sample just before flag j takes effect (float64)."""
    sens = np.float32(-2.0 / N)
    step = (sens * np.asarray(angles, np.float32)).astype(np.float64)
    Phi = np.zeros(len(peaks) + 1)
    for j in range(len(peaks) - 1):
        Phi[j + 1] = Phi[j] + step[j] * float(int(peaks[j + 1]) - int(peaks[j]))
    return Phi, step


def phase_at(idx, peaks, Phi, step):
    idx = np.asarray(idx, np.int64)
    j = np.searchsorted(np.asarray(peaks, np.int64), idx, side="right") - 1
    ph = np.where(j >= 0, Phi[np.maximum(j, 0)] + step[np.maximum(j, 0)] * (idx - np.asarray(peaks, np.int64)[np.maximum(j, 0)] + 1), 0.0)
    return ph


def rx_symbols(cfg, y, peaks, angles, frames):
    """sigmix + sampler + FFT(shift) for every emitted symbol: (flag, Y[N]) list."""
    N, CP = cfg.fft_length, cfg.cp_length
    L = N + CP
    Phi, step = nco_phase(peaks, angles, N)
    out = []
    for p, K in frames:
        for k in range(K + 1):
            s0 = p + k * L - N + 1
            idx = np.arange(s0, s0 + N)
            ph = phase_at(idx, peaks, Phi, step)
            w = y[idx] * np.exp(1j * ph)
            Y = np.fft.fftshift(np.fft.fft(w))
            out.append((1 if k == 0 else 0, Y))
    return out


def frame_acq(cfg, symbols):
    N, occ, CP, const, ks, _ = cfg_arrays(cfg)
    zl = config.zeros_on_left(N, occ)
    sh = cfg.max_fft_shift_len
    kd = np.zeros(occ)
    for i in range(0, occ - 2, 2):
        kd[i] = abs(ks[i] - ks[i + 2]) ** 2
    hinv = np.zeros(occ, complex)
    d = 0
    count = 1
    out = []
    for flag, Y in symbols:
        if flag:
            count = 1
            sd = np.zeros(N)
            sd[:N - 2] = np.abs(Y[:N - 2] - Y[2:]) ** 2
            best, index = 0.0, 0
            for i in range(zl - sh, zl + sh):
                s = float(np.dot(kd, sd[i:i + occ]))
                if s > best:
                    best, index = s, i
            d = index - zl
            comp = np.exp(-2j * np.pi * d * CP / N * 1)
            hinv[0] = ks[0] / (comp * Y[zl + d])
            for i in range(2, occ, 2):
                hinv[i] = ks[i] / (comp * Y[i + zl + d])
                hinv[i - 1] = (hinv[i] + hinv[i - 2]) / 2
            if occ % 2 == 0:
                hinv[occ - 1] = hinv[occ - 2]
        comp = np.exp(-2j * np.pi * d * CP / N * count)
        out.append((flag, hinv * comp * Y[zl + d:zl + d + occ]))
        count += 1
        if count == 1000:
            count = 1
    return out


def frame_sink(cfg, acq):
    """Returns (messages, derotated-symbol list)."""
    N, occ, CP, const, ks, _ = cfg_arrays(cfg)
    nb = nbits_of(cfg)
    smap = np.array(config.carrier_map(occ, occ))
    msgs, derots = [], []
    state = 0
    for flag, sym in acq:
        if state == 0:
            if flag:
                state = 1
                bitbuf = []
                header_bytes = []
                phase = freq = 0.0
                dfe = np.ones(len(smap), complex)
                pkt = bytearray()
            continue
        # demapper (vectorised over carriers; decisions are carrier-local)
        s = sym[smap] * np.exp(1j * phase) * dfe
        derots.append(s)
        dist = np.abs(s[:, None] - const[None, :]) ** 2
        bits = np.argmin(dist, axis=1)
        closest = const[bits]
        acc = np.sum(s * np.conj(closest))
        upd = np.abs(s) ** 2 > 0.001
        dfe = np.where(upd, dfe + cfg.eq_gain * (closest / np.where(upd, s, 1) - dfe), dfe)
        ang = math.atan2(acc.imag, acc.real)
        freq = freq - cfg.freq_gain * ang
        phase = phase + freq - cfg.phase_gain * ang
        if phase >= 2 * math.pi:
            phase -= 2 * math.pi
        if phase < 0:
            phase += 2 * math.pi
        for b in bits:
            for k in range(nb):
                bitbuf.append((int(b) >> k) & 1)
        nbytes = len(bitbuf) // 8
        newbytes = [sum(bitbuf[8 * i + k] << k for k in range(8)) for i in range(nbytes)]
        bitbuf = bitbuf[8 * nbytes:]
        for byte in newbytes:
            if state == 1:
                header_bytes.append(byte)
                if len(header_bytes) == 4:
                    h = int.from_bytes(bytes(header_bytes), "big")
                    if (h >> 16) == (h & 0xFFFF):
                        state = 2
                        plen = (h >> 16) & 0x0FFF
                        pkt = bytearray()
                        if plen == 0:
                            msgs.append(bytes(pkt))
                            state = 0
                    else:
                        state = 0
            elif state == 2:
                pkt.append(byte)
                if len(pkt) == plen:
                    msgs.append(bytes(pkt))
                    state = 0
            # state 0: remaining bytes of the symbol are dropped
    return msgs, derots


def rx(cfg, iq):
    """Full chain in float64.  Returns dict of stage outputs."""
    y = chan_filter(cfg, np.asarray(iq))
    u, P = sync_metric(cfg, y)
    peaks = peak_detect(u, cfg.peak_rise, cfg.peak_fall, cfg.peak_alpha)
    angles = np.angle(P[peaks]) if len(peaks) else np.zeros(0)
    frames = sampler_frames(peaks, len(y), cfg.fft_length, cfg.cp_length, cfg.sampler_timeout)
    syms = rx_symbols(cfg, y, peaks, angles, frames)
    acq = frame_acq(cfg, syms)
    msgs, derots = frame_sink(cfg, acq)
    packets = [ofdm_packet_utils.unmake_packet(m) for m in msgs]
    return dict(y=y, u=u, peaks=peaks, angles=angles, frames=frames, fft=syms, acq=acq, sink=derots,
                msgs=msgs, packets=packets)
