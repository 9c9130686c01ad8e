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
// modulator.  WG = max(256, N/8) threads = SPW symbols of N/8 threads each.
// ---------------------------------------------------------------------------------
#ifndef TX_MIN_WG
#define TX_MIN_WG 256  // (64 = one symbol per workgroup at N = 512 measured the same: the kernel is bound by the noise generator)
#endif
template <int N>
struct TxGeom {
  static constexpr int T = N / 8;
  static constexpr int WG = (T > TX_MIN_WG) ? T : TX_MIN_WG;
  static constexpr int SPW = WG / T;
};

template <int N>
__global__ void __launch_bounds__(TxGeom<N>::WG)
    k_tx_mod(TxParams p, const uint8_t* __restrict__ framed, const uint64_t* __restrict__ framed_off,
             const uint64_t* __restrict__ sym_off, const uint32_t* __restrict__ sym_pkt, uint32_t uniform_spp,
             uint64_t nsym, uint64_t lead, c32* __restrict__ out, c32* __restrict__ freq_tap) {
  constexpr int T = TxGeom<N>::T, SPW = TxGeom<N>::SPW;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  c32* lds = reinterpret_cast<c32*>(smem_raw) + (threadIdx.x / T) * (fft_lds_bufs(N) * fft_lds_points(N));
  const int t = threadIdx.x % T;
  uint64_t sym = (uint64_t)blockIdx.x * SPW + threadIdx.x / T;
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

  c32 e[8];
  if (s == 0) {
    // ofdm_insert_preamble: the known symbol goes out ahead of the packet's first symbol
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = p.preamble[(t + m * T + N / 2) & (N - 1)];
  } else {
    const uint8_t* msg = framed + framed_off[pkt];
    const uint32_t mlen = (uint32_t)(framed_off[pkt + 1] - framed_off[pkt]);
    // bit positions fit 32 bits: a packet holds at most 4 105 bytes
    const uint32_t msgbits = 8u * mlen;
    const uint32_t nb = (uint32_t)p.nbits, bmask = (1u << nb) - 1u;
    const uint32_t bit0 = (s - 1) * (uint32_t)p.nc * nb;  // first message bit of this symbol (may pass the end)
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = (t + m * T + N / 2) & (N - 1);  // ifftshift folded into the index
      const int car = p.bin2car[k];
      c32 v = mk(0.0f, 0.0f);
      if (car >= 0) {
        const uint32_t b0 = bit0 + (uint32_t)car * nb;
        uint32_t bits;
        if (bit0 <= msgbits && b0 + nb <= msgbits) {
          // LSB-first bit stream cut into nbits chunks (digital_ofdm_mapper_bcv::work)
          const uint32_t byte = b0 >> 3;
          uint32_t w = msg[byte];
          if (byte + 1 < mlen) w |= (uint32_t)msg[byte + 1] << 8;
          bits = (w >> (b0 & 7)) & bmask;
        } else {
          const uint64_t slot = (uint64_t)(s - 1) * (uint64_t)p.nc + (uint64_t)car;
          bits = pad_symbol_hash(p.pad_seed, pkt, slot, (uint32_t)p.arity);  // rand() % arity stand-in
        }
        v = p.constellation[bits];
      }
      e[m] = v;
    }
  }
  if (freq_tap && active) {
#pragma unroll
    for (int m = 0; m < 8; m++) freq_tap[sym * N + ((t + m * T + N / 2) & (N - 1))] = e[m];
  }

  fft_run<N, true>(e, t, lds, p.tw, [] { __syncthreads(); });

  if (!active) return;
  const uint64_t base = lead + sym * (uint64_t)p.L;
  c32* o = out + base;
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int n = t + m * T;
    c32 v = e[m];
    v.re = v.re * p.scale1;
    v.im = v.im * p.scale1;
    v.re = v.re * p.amp;
    v.im = v.im * p.amp;
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
