// tx.h -- transmit-side kernels.
//   k_frame_pack : batched make_packet (ofdm_packet_utils.py:99-143): CRC-32, header,
//                  0x55 tail/pad, whitening.
//   k_tx_mod     : ofdm_mapper_bcv + ofdm_insert_preamble + fft_vcc(inverse, shift) +
//                  ofdm_cyclic_prefixer + the two multiply_const_cc (ofdm.py:106-118,
//                  transmit_path.py:48-54), one OFDM symbol per N/8 threads, with the
//                  synthetic channel optionally fused into the store.
//   k_channel    : the same channel on an existing buffer / on noise-only regions.
#pragma once
#include "common.h"
#include "fft.h"
#ifndef TX_PK
#define TX_PK true  // hand-packed butterflies (fft.h)
#endif

struct TxParams {
  int N, CP, L, occ, nc, nbits, arity, zl;
  float scale1;   // 1/sqrt(N)            (ofdm.py:114)
  float amp;      // tx_amplitude         (transmit_path.py:48-54)
  uint64_t pad_seed;
  uint32_t whitener_offset;
  uint32_t pad_for_usrp;
  // tables (device)
  const c32* constellation;   // [arity]
  const c32* preamble;        // [N]   padded known symbol (ofdm.py:83-87)
  const int16_t* bin2car;     // [N]   FFT bin -> data carrier ordinal, -1 if unused
  const c32* tw;              // [N]   exp(-2 pi i k / N)
  const uint8_t* mask;        // [4096]
  const uint32_t* crc_table;  // [256]
  // channel
  int chan_on;
  float sigma, cfo;
  uint64_t seed, stream;
};

// ---------------------------------------------------------------------------------
// make_packet, one WAVE per packet (ofdm_packet_utils.py:99-143).
//   pass 1  CRC-32 of the payload: each lane check-sums 16 bytes of every KiB, pieces combined with
//           crc(A||B) = crc(A) x^(8|B|) + crc(B) (mod P) -- crc_multmodp and the x^(8k) table;
//   pass 2  header | (payload | CRC big-endian | 0x55 tail and pad) XOR mask, composed a KiB at a time in an
//           LDS line and written with coalesced, dword-aligned stores.
// Both passes fetch the payload as coalesced ALIGNED dwords into LDS and realign there (packets start at
// arbitrary byte offsets); a thread-per-packet version wrote 5.4x the bytes it produced (byte stores 1 035
// bytes apart across lanes).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void fp_load_kib(const uint8_t* src, uint32_t c0, uint32_t len, uint32_t* line, int lane) {
  // bytes [c0, min(c0 + 1024, len)) of src -> line[] such that byte i of the piece is byte (sh + i) of line
  const uintptr_t a = reinterpret_cast<uintptr_t>(src + c0);
  const uint32_t* base = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
  const uint32_t sh = (uint32_t)(a & 3u);
  const uint32_t nbytes = (len - c0 < 1024u) ? (len - c0) : 1024u;
  const uint32_t ndw = (sh + nbytes + 3u) >> 2;  // <= 257: never past the dword that holds the last payload byte
  for (uint32_t d = (uint32_t)lane; d < ndw; d += WAVE) line[d] = base[d];
}
// 4 payload bytes starting at piece-relative byte i (line filled by fp_load_kib with shift sh)
__device__ __forceinline__ uint32_t fp_word(const uint32_t* line, uint32_t sh, uint32_t i) {
  const uint32_t q = sh + i;
  return __builtin_amdgcn_alignbyte(line[(q >> 2) + 1], line[q >> 2], q & 3u);
}

__global__ void __launch_bounds__(256) k_frame_pack(TxParams p, const uint8_t* __restrict__ payloads,
                                                     const uint64_t* __restrict__ payload_off,
                                                     const uint32_t* __restrict__ payload_len,
                                                     const uint64_t* __restrict__ framed_off, int npkt,
                                                     uint8_t* __restrict__ framed, const uint32_t* __restrict__ xp8) {
  __shared__ uint32_t tab[256];
  __shared__ __align__(16) uint32_t in_all[4][264];
  __shared__ __align__(16) uint32_t out_all[4][264];
  tab[threadIdx.x] = p.crc_table[threadIdx.x];
  __syncthreads();
  const int lane = lane_id(), w = wave_id();
  uint32_t* in = in_all[w];
  uint32_t* stage = out_all[w];
  const int k = blockIdx.x * 4 + w;
  if (k >= npkt) return;
  const uint8_t* src = payloads + payload_off[k];
  const uint32_t len = payload_len[k];
  uint8_t* out = framed + framed_off[k];
  const uint32_t total = (uint32_t)(framed_off[k + 1] - framed_off[k]);  // 4 + len + 4 + 1 (+ pad)
  const uint32_t off = p.whitener_offset;

  // ---- pass 1: CRC-32 of the payload -----------------------------------------------------------
  uint32_t acc = 0;
  for (uint32_t c0 = 0; c0 < len; c0 += 1024) {
    fp_load_kib(src, c0, len, in, lane);
    if (lane == 0) in[263] = 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src + c0) & 3u);
    const uint32_t o = c0 + 16u * (uint32_t)lane;
    if (o < len) {
      const uint32_t nb = (len - o < 16u) ? (len - o) : 16u;
      uint32_t crc = 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t wd = (4u * (uint32_t)q < nb) ? fp_word(in, sh, 16u * (uint32_t)lane + 4u * (uint32_t)q) : 0u;
#pragma unroll
        for (int b = 0; b < 4; b++)
          if ((uint32_t)(4 * q + b) < nb) crc = tab[(crc ^ (wd >> (8 * b))) & 0xFF] ^ (crc >> 8);
      }
      crc ^= 0xFFFFFFFFu;
      acc ^= crc_multmodp(xp8[len - (o + nb)], crc);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, WAVE);
  const uint32_t crc32 = acc;  // crc32 of an empty payload is 0 = the empty XOR

  // ---- pass 2: the framed packet, a KiB of OUTPUT at a time ---------------------------------------
  const uint32_t L = len + 4;
  const uint32_t val = ((off & 0xF) << 12) | (L & 0x0FFF);  // make_header (ofdm_packet_utils.py:93-97)
  const uint32_t hdr = ((val >> 8) & 0xFF) | ((val & 0xFF) << 8) | (((val >> 8) & 0xFF) << 16) | ((val & 0xFF) << 24);
  for (uint32_t c0 = 0; c0 < total; c0 += 1024) {
    // output bytes [c0, c0+1024) hold body bytes [c0-4, c0+1020): fetch the payload part of that range
    const uint32_t b0 = (c0 >= 4) ? c0 - 4 : 0;  // first body byte this round needs
    if (b0 < len) {
      // (load from a dword-aligned payload position so that fp_word's index math stays simple)
      fp_load_kib(src, b0 & ~3u, len, in, lane);
    }
    if (lane == 0) in[263] = 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const uint32_t pb = b0 & ~3u;  // payload byte at piece-relative 0
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src + pb) & 3u);
    uint32_t wds[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t qo = c0 + 16u * (uint32_t)lane + 4u * (uint32_t)q;  // output byte of this word
      uint32_t wd = 0;
      if (qo < total) {
        if (qo == 0) {
          wd = hdr;
        } else {
          const uint32_t i0 = qo - 4;  // body byte of the word's first byte (a multiple of 4)
          uint32_t body;
          if (i0 + 4 <= len) {
            body = fp_word(in, sh, i0 - pb);
          } else {
            body = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const uint32_t i = i0 + (uint32_t)b;
              uint32_t by;
              if (i < len) by = (fp_word(in, sh, (i & ~3u) - pb) >> (8 * (i & 3u))) & 0xFF;
              else if (i < len + 4) by = (crc32 >> (8 * (3 - (i - len)))) & 0xFF;  // struct.pack(">I", crc)
              else by = 0x55;                                                        // tail + USRP pad
              body |= by << (8 * b);
            }
          }
          // whitening mask bytes off+i0 .. off+i0+3
          const uint32_t mi = off + i0;
          const uint32_t* m32 = reinterpret_cast<const uint32_t*>(p.mask);
          uint32_t mk4;
          if ((mi & 3u) == 0 && mi + 4 <= OFDM_MASK_LEN) {
            mk4 = m32[mi >> 2];
          } else {
            mk4 = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
              if (mi + (uint32_t)b < OFDM_MASK_LEN && i0 + (uint32_t)b < total - 4) mk4 |= (uint32_t)p.mask[mi + b] << (8 * b);
          }
          wd = body ^ mk4;
        }
      }
      wds[q] = wd;
    }
    reinterpret_cast<uint4*>(stage)[lane] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
    if (lane == 0) stage[256] = 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the staging line is written
    const uint32_t rem = (total - c0 < 1024u) ? (total - c0) : 1024u;
    uint8_t* og = out + c0;
    const uint32_t head0 = (4u - (uint32_t)((uintptr_t)og & 3u)) & 3u;
    const uint32_t head = head0 < rem ? head0 : rem;
    const uint32_t nd = (rem - head) >> 2;
    const uint8_t* st8 = reinterpret_cast<const uint8_t*>(stage);
    if ((uint32_t)lane < head) og[lane] = st8[lane];
    for (uint32_t dw = (uint32_t)lane; dw < nd; dw += WAVE) {
      const uint32_t i0 = head + 4u * dw;
      const uint32_t w0 = stage[i0 >> 2], w1 = stage[(i0 >> 2) + 1];
      reinterpret_cast<uint32_t*>(og + i0)[0] = __builtin_amdgcn_alignbyte(w1, w0, i0 & 3u);
    }
    const uint32_t tail0 = head + 4u * nd;
    if (tail0 + (uint32_t)lane < rem) og[tail0 + lane] = st8[tail0 + lane];
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------
// symbol -> (packet, symbol-in-packet) descriptors for ragged batches
// ---------------------------------------------------------------------------------
__global__ void k_sym_desc(const uint64_t* __restrict__ sym_off, int npkt, uint32_t* __restrict__ sym_pkt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npkt) return;
  for (uint64_t s = sym_off[k]; s < sym_off[k + 1]; s++) sym_pkt[s] = (uint32_t)k;
}

// ---------------------------------------------------------------------------------
// modulator.  WG = max(512, N/8) threads = SPW symbols of N/8 threads each.
// ---------------------------------------------------------------------------------
#ifndef TX_MIN_WG
#define TX_MIN_WG 512
#endif
// A wave-sized symbol transforms in ONE LDS buffer (fft_run1: in-place middle passes, wave-level fences), and the
// workgroup is 512 threads: eight such symbols, 39 KB.  Measured at C2 on one box, alternating:
//   256 threads, two buffers (39 KB, 4 waves/SIMD)   k_tx_mod 2.62 ms   pipelined step 13.35-13.48 ms
//   256 threads, one buffer  (20 KB, 7 waves/SIMD)   k_tx_mod 2.30 ms   pipelined step 13.9-14.0 ms
//   512 threads, one buffer  (39 KB, 6 waves/SIMD)   k_tx_mod 2.41 ms   pipelined step 13.30 ms
// The kernel is latency-bound (one short-lived wave per symbol: a chain of dependent loads, then the transform), so it
// wants waves; but 20 KB workgroups squeeze onto compute units already full of the receiver's k_sync workgroups and
// slow those barrier-coupled waves more than the modulator gains, while 39 KB ones wait for a unit to drain.
#ifndef TX_ONEBUF
#define TX_ONEBUF 1
#endif
template <int N>
struct TxGeom {
  static constexpr int T = N / 8;
  static constexpr int WG = (T > TX_MIN_WG) ? T : TX_MIN_WG;
  static constexpr int SPW = WG / T;
  static constexpr bool ONEBUF = (TX_ONEBUF && T <= WAVE) || fft_onebuf(N);
  static constexpr int SYM_POINTS = (ONEBUF ? 1 : 2) * fft_lds_points(N);  // c32 per symbol
  // wave-sized symbols read their twiddles from a table in LDS, staged once per workgroup (fft.h FftTwLds)
#ifndef TX_TW_LDS_BIG
#define TX_TW_LDS_BIG 1
#endif
  static constexpr bool TW_LDS = ONEBUF && (T <= WAVE || TX_TW_LDS_BIG);
  static constexpr int TW_POINTS = TW_LDS ? fft_tw_lds_points(N) : 0;
  static constexpr int lds_bytes() {
    return (SPW * SYM_POINTS + TW_POINTS) * (int)sizeof(c32) + OFDM_MAX_ARITY * (int)sizeof(c32);
  }
};

#ifndef TX_WAVES
#define TX_WAVES 1  // minimum waves per SIMD the register allocation must admit (1: no constraint)
#endif
template <int N, bool LEAN>
__global__ void __launch_bounds__(TxGeom<N>::WG, TX_WAVES)
    k_tx_mod(TxParams p, const uint8_t* __restrict__ framed, const uint64_t* __restrict__ framed_off,
             const uint64_t* __restrict__ sym_off, const uint32_t* __restrict__ sym_pkt, uint32_t uniform_spp,
             uint64_t nsym, uint64_t lead, c32* __restrict__ out, c32* __restrict__ freq_tap,
             c32* __restrict__ ifft_tap) {
  constexpr int T = TxGeom<N>::T, SPW = TxGeom<N>::SPW;
  // LEAN: the kernel of a batch without transmit-side taps and without a carrier offset in the synthetic channel (the
  // benchmark's case).  Both taps and the float64 rotation are then compile-time nothing: 74 -> 54 registers, eight
  // waves per SIMD instead of six (k_tx_mod 2.15 -> 2.08 ms at C2; the pipelined step 11.10 -> 10.98 ms).
  if constexpr (LEAN) {
    freq_tap = nullptr;
    ifft_tap = nullptr;
    p.cfo = 0.0f;
  }
  extern __shared__ __align__(16) unsigned char smem_raw[];
  c32* lds = reinterpret_cast<c32*>(smem_raw) + (threadIdx.x / T) * TxGeom<N>::SYM_POINTS;
  const int t = threadIdx.x % T;
  // A symbol whose threads fill whole waves is the same for every lane of a wave: saying so (readfirstlane) moves the
  // symbol's index arithmetic -- packet, bit offsets, message and output addresses -- from the vector to the scalar unit.
  uint32_t slot = threadIdx.x / T;
  if constexpr (T >= WAVE) slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  uint64_t sym = (uint64_t)blockIdx.x * SPW + slot;
  const bool active = sym < nsym;
  if (!active) sym = nsym - 1;

  uint32_t pkt, s;
  if (uniform_spp) {
    if (nsym <= 0xFFFFFFFFull) {  // (always, in practice) 32-bit divide: a quarter of the 64-bit sequence
      const uint32_t s32 = (uint32_t)sym;
      pkt = s32 / uniform_spp;
      s = s32 - pkt * uniform_spp;
    } else {
      pkt = (uint32_t)(sym / uniform_spp);
      s = (uint32_t)(sym % uniform_spp);
    }
  } else {
    pkt = sym_pkt[sym];
    s = (uint32_t)(sym - sym_off[pkt]);
  }
  if constexpr (T >= WAVE) {
    pkt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pkt);
    s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
  }

  // ---- digital_ofdm_mapper_bcv::work for this symbol ---------------------------------------------------------
  // The symbol's slice of the framed packet (nc * nbits bits) is fetched once, as aligned dwords, into the LDS the
  // transform will use later, and the constellation sits in LDS too: each point then costs two LDS reads instead of
  // a chain of three dependent global loads.
  c32* twl = reinterpret_cast<c32*>(smem_raw) + SPW * TxGeom<N>::SYM_POINTS;  // twiddle table, shared by the workgroup
  c32* cst = twl + TxGeom<N>::TW_POINTS;                                       // [arity], shared by the workgroup
  // The tables' loads are issued here and their values parked in registers; they go to LDS further down, after the
  // message slice's loads have been issued too: the wave waits for ONE round trip to memory, not for one per table
  // (a wave lives for one symbol: this prologue is most of its life).
  constexpr int CSTN = (OFDM_MAX_ARITY + TxGeom<N>::WG - 1) / TxGeom<N>::WG;
  constexpr int TWN = TxGeom<N>::TW_LDS ? (fft_tw_used(N) + TxGeom<N>::WG - 1) / TxGeom<N>::WG : 0;
  c32 cst_v[CSTN], tw_v[TWN > 0 ? TWN : 1];
#pragma unroll
  for (int r = 0; r < CSTN; r++) {
    const int i = (int)threadIdx.x + r * TxGeom<N>::WG;
    cst_v[r] = p.constellation[i < p.arity ? i : p.arity - 1];  // (clamped, not predicated: a predicated load is waited for at once)
  }
#pragma unroll
  for (int r = 0; r < TWN; r++) {
    const int i = (int)threadIdx.x + r * TxGeom<N>::WG;
    tw_v[r] = p.tw[i < fft_tw_used(N) ? i : fft_tw_used(N) - 1];
  }
  // (the carrier of every bin this thread transforms: independent of the message, so in flight with it)
  int car[8];
#pragma unroll
  for (int m = 0; m < 8; m++) car[m] = p.bin2car[(t + m * T + N / 2) & (N - 1)];  // ifftshift folded into the index
  uint32_t* mbytes = reinterpret_cast<uint32_t*>(lds);  // this symbol's message bytes (<= N + 8 of them)
  const uint32_t nb = (uint32_t)p.nbits, bmask = (1u << nb) - 1u;
  const uint8_t* msg = framed + framed_off[pkt];
  const uint32_t mlen = (uint32_t)(framed_off[pkt + 1] - framed_off[pkt]);
  const uint32_t msgbits = 8u * mlen;  // (bit positions fit 32 bits: a packet holds at most 4 105 bytes)
  const uint32_t bit0 = (s == 0) ? 0u : (s - 1) * (uint32_t)p.nc * nb;  // first message bit of this symbol (may pass the end)
  const uint32_t byte0 = bit0 >> 3;
  uint32_t sh = 0;  // the chunk's first byte sits at byte `sh` of mbytes
  if (s != 0 && byte0 < mlen) {
    uint32_t bend = ((bit0 + (uint32_t)p.nc * nb + 7u) >> 3) + 1u;  // one byte past the last chunk's straddle
    if (bend > mlen) bend = mlen;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(msg + byte0);
    const uint32_t* base = reinterpret_cast<const uint32_t*>(a0 & ~(uintptr_t)3);
    sh = (uint32_t)(a0 & 3u);
    const uint32_t ndw = (sh + (bend - byte0) + 3u) >> 2;
    for (uint32_t d = (uint32_t)t; d < ndw; d += T) mbytes[d] = base[d];
  }
#pragma unroll
  for (int r = 0; r < CSTN; r++) {
    const int i = (int)threadIdx.x + r * TxGeom<N>::WG;
    if (i < p.arity) cst[i] = cst_v[r];
  }
#pragma unroll
  for (int r = 0; r < TWN; r++) {
    const int i = (int)threadIdx.x + r * TxGeom<N>::WG;
    if (i < fft_tw_used(N)) twl[lpad(i)] = tw_v[r];
  }
  __syncthreads();  // (the tables are shared by the workgroup's symbols)
  c32 e[8];
  if (s == 0) {
    // ofdm_insert_preamble: the known symbol goes out ahead of the packet's first symbol
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = p.preamble[(t + m * T + N / 2) & (N - 1)];
  } else {
    const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(mbytes);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      c32 v = mk(0.0f, 0.0f);
      if (car[m] >= 0) {
        const uint32_t b0 = bit0 + (uint32_t)car[m] * nb;
        uint32_t bits;
        if (bit0 <= msgbits && b0 + nb <= msgbits) {
          // LSB-first bit stream cut into nbits chunks (digital_ofdm_mapper_bcv::work)
          const uint32_t byte = b0 >> 3, li = sh + (byte - byte0);
          uint32_t w = mb8[li];
          if (byte + 1 < mlen) w |= (uint32_t)mb8[li + 1] << 8;
          bits = (w >> (b0 & 7)) & bmask;
        } else {
          const uint64_t slot = (uint64_t)(s - 1) * (uint64_t)p.nc + (uint64_t)car[m];
          bits = pad_symbol_hash(p.pad_seed, pkt, slot, (uint32_t)p.arity);  // rand() % arity stand-in
        }
        v = cst[bits];
      }
      e[m] = v;
    }
  }
  // the symbol's threads are done with its message bytes: the transform may use the LDS
  if constexpr (T <= WAVE) {
    FftWaveSync()();
  } else {
    __syncthreads();
  }
  if (freq_tap && active) {
#pragma unroll
    for (int m = 0; m < 8; m++) freq_tap[sym * N + ((t + m * T + N / 2) & (N - 1))] = e[m];
  }

  // up to N = 512 a symbol's N/8 threads sit inside one wave: no workgroup barrier in the exchanges
  if constexpr (T <= WAVE) {
    if constexpr (TxGeom<N>::TW_LDS)
      fft_run1<N, true, TX_PK, FftWaveSync, FftTwLds>(e, t, lds, FftTwLds{twl}, FftWaveSync());
    else if constexpr (TxGeom<N>::ONEBUF)
      fft_run1<N, true, TX_PK, FftWaveSync, FftTwTable>(e, t, lds, FftTwTable{p.tw}, FftWaveSync());
    else
      fft_run<N, true, FftWaveSync, TX_PK>(e, t, lds, p.tw, FftWaveSync());
  } else {
    if constexpr (TxGeom<N>::TW_LDS)
      fft_run_tw<N, true, FftBlockSync, TX_PK, FftTwLds>(e, t, lds, FftTwLds{twl}, FftBlockSync());
    else
      fft_run<N, true, FftBlockSync, TX_PK>(e, t, lds, p.tw, FftBlockSync());
  }

  if (!active) return;
  if (ifft_tap) {  // ofdm_ifft_c.dat (ofdm.py:128): the transform's output vectors, before the cyclic prefix
#pragma unroll
    for (int m = 0; m < 8; m++) ifft_tap[sym * N + (uint64_t)(t + m * T)] = e[m];
  }
  const uint64_t base = lead + sym * (uint64_t)p.L;
  c32* o = out + base;
  c32 v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    v[m] = e[m];
    v[m].re = v[m].re * p.scale1;
    v[m].im = v[m].im * p.scale1;
    v[m].re = v[m].re * p.amp;
    v[m].im = v[m].im * p.amp;
  }
  if (!p.chan_on) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int n = t + m * T;
      o[p.CP + n] = v[m];
      if (n >= N - p.CP) o[n - (N - p.CP)] = v[m];  // ofdm_cyclic_prefixer: out[0:CP] = in[N-CP:N]
    }
    return;
  }
  // ---- synthetic channel fused into the store -------------------------------------------------------------
  // Noise words come in pairs (even, odd stream position).  When the symbol starts on an even position and CP is
  // even, lanes t and t^1 hold the two samples of a pair for every m: the even lane draws the pairs of the even
  // m, the odd lane those of the odd m, and they swap the halves they do not need (one DPP move).
  const uint32_t key = chan_key(p.seed, p.stream);
  const bool paired = p.sigma > 0.0f && (((base + (uint64_t)p.CP) | (uint64_t)p.CP) & 1ull) == 0;
  uint32_t wb[8], wc[8];  // noise words of the body samples / of their cyclic-prefix copies
  if (paired) {
    const int odd = t & 1;
#pragma unroll
    for (int mm = 0; mm < 4; mm++) {
      const int mine = 2 * mm + odd;
      const int n = t + mine * T;
      uint32_t we, wo;
      chan_pair_words(base + (uint64_t)(p.CP + n), key, we, wo);
      const uint32_t keep = odd ? wo : we, give = odd ? we : wo;
      const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
      wb[2 * mm] = odd ? got : keep;
      wb[2 * mm + 1] = odd ? keep : got;
      uint32_t ce = 0, co = 0;
      if (n >= N - p.CP) chan_pair_words(base + (uint64_t)(n - (N - p.CP)), key, ce, co);
      const uint32_t ckeep = odd ? co : ce, cgive = odd ? ce : co;
      const uint32_t cgot = (uint32_t)__builtin_amdgcn_mov_dpp((int)cgive, 0xB1, 0xF, 0xF, true);
      wc[2 * mm] = odd ? cgot : ckeep;
      wc[2 * mm + 1] = odd ? ckeep : cgot;
    }
  }
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int n = t + m * T;
    const int pos = p.CP + n;
    c32 a = v[m];
    if (paired) {
      if (p.cfo != 0.0f) a = chan_rotate(a, base + (uint64_t)pos, p.cfo);
      a = chan_add_noise(a, wb[m], p.sigma);
    } else {
      a = channel_apply(v[m], base + (uint64_t)pos, p.sigma, p.cfo, p.seed, p.stream);
    }
    o[pos] = a;
    if (n >= N - p.CP) {  // ofdm_cyclic_prefixer: out[0:CP] = in[N-CP:N]
      const int pc = n - (N - p.CP);
      c32 b = v[m];
      if (paired) {
        if (p.cfo != 0.0f) b = chan_rotate(b, base + (uint64_t)pc, p.cfo);
        b = chan_add_noise(b, wc[m], p.sigma);
      } else {
        b = channel_apply(v[m], base + (uint64_t)pc, p.sigma, p.cfo, p.seed, p.stream);
      }
      o[pc] = b;
    }
  }
}

// ---------------------------------------------------------------------------------
// channel on a buffer (in place) or noise-only fill (x == 0) for lead-in / tail
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_channel(c32* __restrict__ iq, uint64_t n, uint64_t index0, int zero_input,
                                                  float sigma, float cfo, uint64_t seed, uint64_t stream) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    c32 x = zero_input ? mk(0.0f, 0.0f) : iq[i];
    iq[i] = channel_apply(x, index0 + i, sigma, cfo, seed, stream);
  }
}
