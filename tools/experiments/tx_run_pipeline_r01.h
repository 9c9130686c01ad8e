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
// make_packet, one thread per packet.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_frame_pack(TxParams p, const uint8_t* __restrict__ payloads,
                                                     const uint64_t* __restrict__ payload_off,
                                                     const uint32_t* __restrict__ payload_len,
                                                     const uint64_t* __restrict__ framed_off, int npkt,
                                                     uint8_t* __restrict__ framed) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = p.crc_table[threadIdx.x];
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npkt) return;
  const uint8_t* src = payloads + payload_off[k];
  const uint32_t len = payload_len[k];
  uint8_t* out = framed + framed_off[k];
  const uint32_t total = (uint32_t)(framed_off[k + 1] - framed_off[k]);
  const uint32_t L = len + 4;
  const uint32_t off = p.whitener_offset;
  const uint32_t val = ((off & 0xF) << 12) | (L & 0x0FFF);  // make_header (ofdm_packet_utils.py:93-97)
  out[0] = (uint8_t)(val >> 8);
  out[1] = (uint8_t)val;
  out[2] = (uint8_t)(val >> 8);
  out[3] = (uint8_t)val;
  uint8_t* body = out + 4;
  uint32_t crc = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < len; i++) {
    const uint8_t b = src[i];
    crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8);
    body[i] = b ^ p.mask[off + i];
  }
  crc ^= 0xFFFFFFFFu;
  body[len + 0] = (uint8_t)(crc >> 24) ^ p.mask[off + len + 0];  // struct.pack(">I", crc)
  body[len + 1] = (uint8_t)(crc >> 16) ^ p.mask[off + len + 1];
  body[len + 2] = (uint8_t)(crc >> 8) ^ p.mask[off + len + 2];
  body[len + 3] = (uint8_t)crc ^ p.mask[off + len + 3];
  for (uint32_t i = L; i < total - 4; i++) body[i] = 0x55 ^ p.mask[off + i];  // tail + USRP pad
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
// modulator.  A group of N/8 threads (one wave at N = 512) walks a run of TX_RUN consecutive
// symbols; a workgroup is max(64, N/8) threads, i.e. one group for N >= 512.
//
// What bounds this kernel is load latency, not arithmetic, so the loop keeps every global load
// off the critical path: the per-thread tables (bin -> carrier of its eight bins, the preamble
// values) are read once per run, the constellation sits in LDS, and the packet bytes of symbol
// i+1 are fetched as coalesced aligned dwords (one or two per lane) while symbol i is being
// transformed; a chunk of nbits is then cut out of two neighbouring LDS dwords with one
// v_alignbit.
// ---------------------------------------------------------------------------------
#define TX_RUN 16
template <int N>
struct TxGeom {
  static constexpr int T = N / 8;                 // threads per symbol
  static constexpr int WG = (T > 64) ? T : 64;    // workgroup
  static constexpr int SPW = WG / T;              // groups (symbol runs) per workgroup
  static constexpr int BITW = N / 4 + 4;          // dwords of packet bytes one symbol can need (+ alignment slack)
  static constexpr int BITR = (BITW + T - 1) / T; // dwords per lane
  // per group: two FFT buffers, two bit buffers; per workgroup: the constellation
  static constexpr size_t lds_bytes() { return (size_t)SPW * (fft_lds_bytes(N) + 2 * BITW * 4) + OFDM_MAX_ARITY * sizeof(c32); }
};

struct TxSymDesc {
  uint32_t pkt, s;        // packet, symbol within the packet (0 = preamble)
  uint32_t mlen;          // framed length of the packet, bytes
  uint32_t bit0;          // first message bit of the symbol
  uint64_t msg_abs;       // byte offset of the packet inside `framed`
  uint64_t dw0;           // first aligned dword (index into framed viewed as uint32) the symbol needs
};

template <int N>
__global__ void __launch_bounds__(TxGeom<N>::WG)
    k_tx_mod(TxParams p, const uint8_t* __restrict__ framed, const uint64_t* __restrict__ framed_off,
             const uint64_t* __restrict__ sym_off, const uint32_t* __restrict__ sym_pkt, uint32_t uniform_spp,
             uint64_t nsym, uint64_t lead, c32* __restrict__ out, c32* __restrict__ freq_tap) {
  using G = TxGeom<N>;
  constexpr int T = G::T, SPW = G::SPW, BITW = G::BITW, BITR = G::BITR;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int grp = threadIdx.x / T, t_ = threadIdx.x % T;
#ifdef TX_HOIST
  const int t = t_;
#endif
  c32* lds = reinterpret_cast<c32*>(smem_raw) + grp * (2 * fft_lds_points(N));
  uint32_t* bitbuf = reinterpret_cast<uint32_t*>(smem_raw + (size_t)SPW * fft_lds_bytes(N)) + grp * (2 * BITW);
  c32* lut = reinterpret_cast<c32*>(smem_raw + (size_t)SPW * (fft_lds_bytes(N) + 2 * BITW * 4));
  const uint32_t* framed32 = reinterpret_cast<const uint32_t*>(framed);  // hipMalloc'ed: 256-byte aligned

  for (int i = threadIdx.x; i < p.arity; i += G::WG) lut[i] = p.constellation[i];

  const uint64_t run0 = ((uint64_t)blockIdx.x * SPW + grp) * TX_RUN;
  // tables of this thread's eight bins
  int car[8];
#pragma unroll
  for (int m = 0; m < 8; m++) car[m] = p.bin2car[(t_ + m * T + N / 2) & (N - 1)];  // ifftshift folded into the index
  const uint32_t symbits = (uint32_t)p.nc * (uint32_t)p.nbits;

  auto describe = [&](uint64_t sym) -> TxSymDesc {
    TxSymDesc d;
    if (uniform_spp) {
      d.pkt = (uint32_t)(sym / uniform_spp);
      d.s = (uint32_t)(sym % uniform_spp);
    } else {
      d.pkt = sym_pkt[sym];
      d.s = (uint32_t)(sym - sym_off[d.pkt]);
    }
    d.msg_abs = framed_off[d.pkt];
    d.mlen = (uint32_t)(framed_off[d.pkt + 1] - d.msg_abs);
    d.bit0 = (d.s ? d.s - 1 : 0) * symbits;
    d.dw0 = (d.msg_abs + (d.bit0 >> 3)) >> 2;
    return d;
  };
  // aligned dwords covering the symbol's bytes (reads at most 3 bytes before the packet and a few
  // past its last needed byte, all inside the framed buffer and its allocation slack)
  auto fetch = [&](const TxSymDesc& d, uint32_t w[BITR]) {
#pragma unroll
    for (int r = 0; r < BITR; r++) {
      const int j = t_ + r * T;
      w[r] = (d.s != 0 && j < BITW) ? framed32[d.dw0 + j] : 0u;
    }
  };

  uint32_t w[BITR];
  TxSymDesc cur, nxt;
  const bool any = run0 < nsym;
  cur = describe(any ? run0 : nsym - 1);
  fetch(cur, w);
  nxt = cur;

#pragma unroll 1
  for (int it = 0; it < TX_RUN; it++) {
    const uint64_t sym = run0 + it;
    const bool active = sym < nsym;
#ifndef TX_HOIST
    // opaque copy of the lane's index, renewed every symbol: keeps the compiler from hoisting every
    // t-dependent address and twiddle of the transform out of the loop (150 VGPRs, 2 waves/SIMD)
    int t = t_;
    asm volatile("" : "+v"(t));
#endif
    uint32_t* bb = bitbuf + (it & 1) * BITW;
#pragma unroll
    for (int r = 0; r < BITR; r++) {
      const int j = t + r * T;
      if (j < BITW) bb[j] = w[r];
    }
    // next symbol's bytes: in flight during this symbol's transform
    if (it + 1 < TX_RUN) {
      nxt = describe((sym + 1 < nsym) ? sym + 1 : nsym - 1);
      fetch(nxt, w);
    }
    __syncthreads();  // bit buffer (and, first time, the constellation) visible; previous transform's LDS reads done

    c32 e[8];
    if (cur.s == 0) {
      // ofdm_insert_preamble: the known symbol goes out ahead of the packet's first symbol
#pragma unroll
      for (int m = 0; m < 8; m++) e[m] = p.preamble[(t + m * T + N / 2) & (N - 1)];
    } else {
      const uint32_t msgbits = 8u * cur.mlen;
      // bit position of the symbol's first chunk relative to the first fetched dword
      const uint32_t rel0 = (uint32_t)(((cur.msg_abs << 3) + cur.bit0) - (cur.dw0 << 5));
#pragma unroll
      for (int m = 0; m < 8; m++) {
        c32 v = mk(0.0f, 0.0f);
        if (car[m] >= 0) {
          const uint32_t cb = (uint32_t)car[m] * (uint32_t)p.nbits;
          uint32_t bits;
          if (cur.bit0 + cb + (uint32_t)p.nbits <= msgbits) {
            // LSB-first bit stream cut into nbits chunks (digital_ofdm_mapper_bcv::work)
            const uint32_t q = rel0 + cb;
            const uint32_t lo = bb[q >> 5], hi = bb[(q >> 5) + 1];
            bits = __builtin_amdgcn_alignbit(hi, lo, q & 31) & ((1u << p.nbits) - 1u);
          } else {
            const uint64_t slot = (uint64_t)(cur.s - 1) * (uint64_t)p.nc + (uint64_t)car[m];
            bits = pad_symbol_hash(p.pad_seed, cur.pkt, slot, (uint32_t)p.arity);  // rand() % arity stand-in
          }
          v = lut[bits];
        }
        e[m] = v;
      }
    }
    if (freq_tap && active) {
#pragma unroll
      for (int m = 0; m < 8; m++) freq_tap[sym * N + ((t + m * T + N / 2) & (N - 1))] = e[m];
    }

    fft_run<N, true>(e, t, lds, p.tw, [] { __syncthreads(); });

    if (active) {
      const uint64_t base = lead + sym * (uint64_t)p.L;
      c32* o = out + base;
      const float sc = p.scale1, amp = p.amp;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int n = t + m * T;
        c32 v = e[m];
        v.re = v.re * sc;   // multiply_const_cc(1/sqrt(N)) then the amp block: two roundings, as the reference
        v.im = v.im * sc;
        v.re = v.re * amp;
        v.im = v.im * amp;
        const int pos = p.CP + n;
        c32 a = v;
        if (p.chan_on) a = channel_apply(v, base + (uint64_t)pos, p.sigma, p.cfo, p.seed, p.stream);
        o[pos] = a;
        if (n >= N - p.CP) {  // ofdm_cyclic_prefixer: out[0:CP] = in[N-CP:N]
          const int pc = n - (N - p.CP);
          c32 b = v;
          if (p.chan_on) b = channel_apply(v, base + (uint64_t)pc, p.sigma, p.cfo, p.seed, p.stream);
          o[pc] = b;
        }
      }
    }
    cur = nxt;
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
