// rx_demod.h -- everything after the timing flags.
//
//   k_frames    closed forms of gr_sample_and_hold_ff + gr_frequency_modulator_fc and of the
//               digital_ofdm_sampler automaton (ofdm_receiver.py~:123-125,133-136): every flag
//               that the sampler can see starts a frame; its number of data symbols depends only
//               on the next flag, the time-out and the end of the stream.
//   k_rx_demod  one workgroup (N/8 threads) per frame: gather N samples past the CP, derotate by
//               the NCO phase, FFT (LDS), frame acquisition (coarse offset + one-tap equaliser),
//               frame sink (PLL + DFE + slicer + bit packing + header parse).
//   chains      a frame sink that is still collecting a packet when the next preamble arrives
//               treats that preamble as data (digital_ofdm_frame_sink::work has no resync in
//               HAVE_SYNC / HAVE_HEADER).  Every frame is demodulated optimistically as if the
//               sink were searching; frames that turn out to be swallowed by an earlier packet are
//               invalidated afterwards (k_chain_*).
//   k_deframe   unmake_packet (ofdm_packet_utils.py:169-191): dewhiten, CRC-32 check, compaction
//               of the (ok, payload) pairs in stream order.
#pragma once
#include "common.h"
#include "fft.h"
#ifndef DEMOD_PK
#define DEMOD_PK true  // hand-packed butterflies (fft.h)
#endif
#include "host_util.h"

#define RAW_SLOT 4096  // bytes of frame-sink message storage per frame (MAX_PKT_LEN)

enum { FR_BAD_HEADER = 0, FR_COMPLETE = 1, FR_INCOMPLETE = 2 };

struct FrameResult {
  uint32_t status;     // FR_*
  uint32_t packetlen;  // header length field (valid when status != BAD_HEADER and header parsed)
  uint32_t end_frame;  // frame index in which the sink went back to SYNC_SEARCH (own index if none)
  uint32_t header_ok;
};

struct FramesParams {
  uint64_t npeaks, nsamples;
  int N, L;
  uint32_t timeout;
  float sens;  // float(-2/N)
  const uint64_t* peaks;
  const c32* peak_P;
  float* angle;     // [npeaks]
  double* step;     // [npeaks]
  uint64_t* inc;    // [npeaks] phase advance until the next flag, in units of 2^-64 turn (see nco_turns)
  uint32_t* K;      // [npeaks] data symbols of the frame
  uint64_t* nsym;   // [npeaks] K+1 for accepted frames else 0 (scanned into symbol ordinals)
  unsigned int* n_lo;   // flags before the sampler's first window (p < N)
  unsigned int* n_hi;   // flags the sampler never reaches (end of stream)
  // chunked streams (ofdm_rx_set_flag_history): the first nforced flags were settled by earlier calls and
  // carry their known phase step instead of one derived from this call's correlator output
  uint64_t nforced;
  const double* forced_step;
  // SYNC "fixed": the NCO's frequency input is a constant, not complex_to_arg(P) at the flag
  int fixed_on;
  float fixed_angle;
};

// NCO phase bookkeeping across flags is done in integers: a phase advance x (radians) becomes
// frac(x / 2 pi) * 2^64 and phases add modulo 2^64 = modulo one turn.  Integer addition is associative, so
// the device-wide scan gives the same phases whatever its shape -- one call on a whole capture and the
// chunked calls of ofdm_demod.feed() agree to the last bit -- and the wrapped phase keeps full precision
// however long the stream runs.
__host__ __device__ __forceinline__ uint64_t nco_turns(double x) {
  double t = x * 0.15915494309189533577;  // 1 / (2 pi)
  t -= floor(t);
  return (uint64_t)(t * 18446744073709551616.0);  // t < 1: exact scaling, no overflow
}
__host__ __device__ __forceinline__ double nco_radians(uint64_t u) {
  return (double)(int64_t)u * 3.4061215800865545e-19;  // 2 pi / 2^64: phase in [-pi, pi)
}

__device__ __forceinline__ uint32_t sampler_K(uint64_t p, bool has_next, uint64_t nxt, uint64_t ns, int L, uint32_t timeout) {
  uint64_t k = (uint64_t)timeout + 1;
  const uint64_t ks = (ns >= p + 2) ? (ns - p - 2) / (uint64_t)L : 0;
  if (ks < k) k = ks;
  if (has_next) {
    const uint64_t kn = (nxt - p - 2) / (uint64_t)L;  // nxt >= p + 1; (nxt-p-2) may wrap for nxt == p+1
    if (nxt >= p + 2) {
      if (kn < k) k = kn;
    } else {
      k = 0;
    }
  }
  return (uint32_t)k;
}

__global__ void __launch_bounds__(256) k_frames(FramesParams q) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= q.npeaks) return;
  const uint64_t p = q.peaks[j];
  const c32 P = q.peak_P[j];
  const float ang = q.fixed_on ? q.fixed_angle : det_atan2f(P.im, P.re);  // complex_to_arg, bit-reproducible form
  q.angle[j] = ang;
  double st = (double)(q.sens * ang);
  if (j < q.nforced) st = q.forced_step[j];
  q.step[j] = st;
  const bool has_next = j + 1 < q.npeaks;
  const uint64_t nxt = has_next ? q.peaks[j + 1] : 0;
  q.inc[j] = has_next ? nco_turns(st * (double)(nxt - p)) : 0ull;

  const uint64_t N = (uint64_t)q.N, L = (uint64_t)q.L;
  bool accept = p >= N;
  if (!accept) atomicAdd(q.n_lo, 1u);
  uint64_t b;  // base of the sampler call that finds this flag
  const bool prev_acc = j > 0 && q.peaks[j - 1] >= N;
  if (prev_acc) {
    const uint64_t pp = q.peaks[j - 1];
    const uint32_t pK = sampler_K(pp, true, p, q.nsamples, q.L, q.timeout);
    const uint64_t bnext = pp - N + 1 + (uint64_t)pK * L;  // base after the previous frame's last symbol
    if (pK == q.timeout + 1) {
      // previous frame timed out: NO_SIG windows of L+1 from there
      const uint64_t m = (p - N - bnext) / (L + 1);
      b = bnext + m * (L + 1);
    } else {
      b = bnext;
    }
  } else {
    b = accept ? ((p - N) / (L + 1)) * (L + 1) : 0;
  }
  if (accept && !(b + L + N < q.nsamples)) {
    accept = false;
    atomicAdd(q.n_hi, 1u);
  }
  const uint32_t K = accept ? sampler_K(p, has_next, nxt, q.nsamples, q.L, q.timeout) : 0;
  q.K[j] = K;
  q.nsym[j] = accept ? (uint64_t)K + 1 : 0;
}

// ------------------------------------------------------------------------------------
// A constellation whose points are every combination of nr real and ni imaginary levels (the QAM tables of
// qam.py:29-73): the slicer then needs the two levels around each part, not a search over the whole table.
struct SlicerGrid {
  int nr, ni;
  float bound;        // beyond it (garbage frames) float32 rounding could rank a far point first: full search
  float lr[16], li[16];  // levels, ascending
  uint8_t idx[256];   // constellation index of (real level a, imaginary level b) at a * ni + b
};

// Frame counts known on the device only: the sampler's bookkeeping (k_frames) leaves the flags it rejects at either end
// of the stream in two counters; kernels queued behind it without a host round trip take j0 / nframes from there
// (lo == nullptr: the host's values are in force).
struct DynFrames {
  const unsigned int* lo;
  const unsigned int* hi;
  uint32_t npeaks;
};
__device__ __forceinline__ uint32_t dyn_nframes(const DynFrames& d, uint32_t stat) { return d.lo ? d.npeaks - *d.lo - *d.hi : stat; }
__device__ __forceinline__ uint32_t dyn_j0(const DynFrames& d, uint32_t stat) { return d.lo ? *d.lo : stat; }

struct DemodParams {
  DynFrames dyn;
  int N, CP, L, occ, zl, nmap, nbits, arity, shift;
  float phase_gain, freq_gain, eq_gain;
  // Sign slicer for the two-level constellations (BPSK on the real axis, QPSK = a symmetric 2 x 2 grid): a point whose
  // parts lie clear of the decision boundaries (sign_eps) and inside sign_bound has its nearest table entry in the
  // quadrant / half-plane of its signs, with a margin far above the float32 rounding of the distances the full
  // search compares -- the full search's answer at two compares.  Points that fail the test take the full search.
  //   sign_kind 0: none   1: index = sign_idx[re > 0]   2: index = sign_idx[2 (re > 0) + (im > 0)]
  int sign_kind;
  unsigned char sign_idx[4];
  float sign_eps, sign_bound;
  uint64_t nsamples;
  uint32_t j0, nframes, npeaks;  // frames are peaks[j0 .. j0+nframes)
  int tap_mode;                  // 0: optimistic pass over all frames; 1: tap pass over valid frames
  const c32* y;
  const uint64_t* peaks;
  const double* Phi;   // [npeaks] NCO phase of the sample just before flag j takes effect
  const double* step;  // [npeaks]
  const uint32_t* K;   // [npeaks]
  const uint64_t* sym_base;  // [npeaks] ordinal of the frame's preamble among emitted symbols
  const c32* tw;
  const c32* ks;        // [occ]
  const float* kd;      // [occ]
  const int16_t* smap;  // [nmap]
  int smap_lds;         // the kernel keeps its own copy of the map in LDS
  const c32* constellation;
  const SlicerGrid* grid;  // nullptr: no grid structure (PSK, small tables): search the table
  const uint8_t* invalid;  // [nframes] (tap pass)
  FrameResult* res;        // [nframes]
  uint8_t* raw;            // [nframes][RAW_SLOT]
  c32* tap_sampler;        // optional [nsym][N]: the FFT's input
  c32* tap_fft;            // optional [nsym][N]
  c32* tap_acq;            // optional [nsym][occ]
  c32* tap_sink;           // optional [nsym][occ]
  uint8_t* tap_demapped;   // optional [nsym]
  // NCO state carried in from the previous chunk: in force before the first flag of this call
  int ref_on;
  int64_t ref_peak;
  double ref_phi, ref_step;
};

// LDS of one frame's workgroup: fft (2 buffers; the first doubles as the shifted spectrum, the second as
// the |Y[i]-Y[i+2]|^2 scratch of the preamble) | hinv[occ] | dfe[occ] | constellation | reduction scratch (more than
// one wave only) | the bits of one OFDM symbol | slicer grid (grid constellations only) | the sink's carrier map
__host__ __device__ inline int demod_symbits_words(int nmap, int nbits) { return (nmap * nbits + 8 + 31) / 32 + 2; }
// The transform works in ONE LDS buffer; the twiddle table sits beside it in LDS (fft.h FftTwLds) -- always up to
// N = 1024, and for longer transforms when that does not cost a workgroup per CU (template parameter TWL, the host
// decides: launch_demod).  hinv doubles as the correlator scratch (occ + 2*shift + 1 floats): it is rebuilt right after.
__host__ __device__ inline int demod_hinv_len(int n, int occ, int shift) {
  const int need = (occ + 2 * shift + 2 + 1) / 2;
  return need > occ ? need : occ;
}
// reduction scratch: per-wave partials beyond one wave, per-thread terms (sequential sum) below one
__host__ __device__ inline int demod_red_floats(int n) { return n / 8 == WAVE ? 0 : 64; }
// The sink's carrier map is read once per carrier and symbol: in LDS too, unless that costs a workgroup per CU
// (DemodParams::smap_lds, the host decides).
// A one-wave frame (N = 512) shares its workgroup with DEMOD_FPW - 1 others: the tables every frame reads (twiddles,
// constellation, slicer grid, carrier map: 4.4 KB at C2) are staged once per workgroup, which brings the LDS a frame
// costs from 12.4 to 9.1 KB -- sixteen frames per CU instead of twelve.  The frames of a workgroup never meet at a
// barrier after the staging (each is a wave of its own: wave-level fences only).
#ifndef DEMOD_FPW
#define DEMOD_FPW 4
#endif
__host__ __device__ constexpr int demod_fpw(int n) { return n / 8 == WAVE ? DEMOD_FPW : 1; }
// tables shared by the frames of a workgroup
__host__ __device__ inline int demod_lds_shared(int n, bool twl, bool smap_lds, int arity, int nmap, bool grid) {
  const int b = (twl ? fft_tw_lds_points(n) : 0) * (int)sizeof(c32) + arity * (int)sizeof(c32) +
                (grid ? (int)sizeof(SlicerGrid) : 0) + (smap_lds ? ((nmap * 2 + 3) & ~3) : 0);
  return (b + 15) & ~15;
}
// one frame's own: transform buffer | hinv | dfe | reduction scratch | the bits of one symbol
__host__ __device__ inline int demod_lds_frame(int n, int occ, int nmap, int nbits, int shift) {
  const int b = fft_lds_points(n) * (int)sizeof(c32) + (demod_hinv_len(n, occ, shift) + occ) * (int)sizeof(c32) +
                demod_red_floats(n) * (int)sizeof(float) + ((demod_symbits_words(nmap, nbits) + 3) & ~3) * 4;
  return (b + 15) & ~15;
}
__host__ __device__ inline int demod_lds_bytes(int n, bool twl, bool smap_lds, int occ, int arity, int nmap, int nbits, int shift,
                                               bool grid) {
  return demod_lds_shared(n, twl, smap_lds, arity, nmap, grid) + demod_fpw(n) * demod_lds_frame(n, occ, nmap, nbits, shift);
}

// Two wave-wide sums in the normative order (oracle: lane_tree_sum -- partner = lane ^ d for d = 32, 16, ... 1, each
// level adding own + partner), without LDS traffic: gfx950's v_permlane32/16_swap exchange half-waves / rows, the
// levels inside a row of 16 are DPP row shifts (the lanes below the partner take the left shift, those above the right
// one: two bank-masked instructions per level) and quad permutations.  Same tree as six ds_bpermute round trips per
// value, at a tenth of their latency.  (Inline asm: two wait states separate a VGPR write from its DPP / swap read.)
__device__ __forceinline__ void wave_sum2_tree(float& a, float& b) {
  float t0, t1;
  asm volatile(
      "v_mov_b32 %2, %0\n\t"
      "v_mov_b32 %3, %1\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %2\n\t"
      "v_permlane32_swap_b32 %1, %3\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32 %1, %1, %3\n\t"
      "v_mov_b32 %2, %0\n\t"
      "v_mov_b32 %3, %1\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_permlane16_swap_b32 %1, %3\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32 %1, %1, %3\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "=&v"(t0), "=&v"(t1));
}

#ifndef DEMOD_TREE_DPP
#define DEMOD_TREE_DPP 1
#endif
// sum over the N/8 threads of the frame (two values at once)
template <int T>
__device__ __forceinline__ void block_sum2_f(float& a, float& b, float* red) {
  if (T >= WAVE) {
#if DEMOD_TREE_DPP
    wave_sum2_tree(a, b);
#else
    a = wave_sum(a);
    b = wave_sum(b);
#endif
    if (T > WAVE) {
      if (lane_id() == 0) {
        red[2 * wave_id()] = a;
        red[2 * wave_id() + 1] = b;
      }
      __syncthreads();
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int i = 0; i < T / WAVE; i++) {
        sa += red[2 * i];
        sb += red[2 * i + 1];
      }
      __syncthreads();
      a = sa;
      b = sb;
    }
  } else {
    red[2 * threadIdx.x] = a;
    red[2 * threadIdx.x + 1] = b;
    __syncthreads();
    float sa = 0.f, sb = 0.f;
    for (int i = 0; i < T; i++) {
      sa += red[2 * i];
      sb += red[2 * i + 1];
    }
    __syncthreads();
    a = sa;
    b = sb;
  }
}

struct dc {  // float64 complex, for the NCO phasor recurrences
  double re, im;
};
__device__ __forceinline__ dc dmul(dc a, dc b) {
  dc r;
  r.re = a.re * b.re - a.im * b.im;
  r.im = a.re * b.im + a.im * b.re;
  return r;
}
__device__ __forceinline__ double f64_uniform(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float f32_uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ dc dc_uniform(dc v) {
  dc r;
  r.re = f64_uniform(v.re);
  r.im = f64_uniform(v.im);
  return r;
}
// exp(j ph) for the NCO, bit-reproducible: wrap to [-pi, pi], reduce by multiples of pi/2 (two-part constant),
// fdlibm kernel polynomials with explicit fma() -- the CPU restatement evaluates the same operations.
__device__ __noinline__ dc dexpj(double ph) {
  ph = ph - 6.283185307179586476925 * floor(ph / 6.283185307179586476925 + 0.5);
  const double kd = rint(ph * 0.63661977236758134308);
  double y = fma(-kd, 1.57079632673412561417e+00, ph);
  y = fma(-kd, 6.07710050650619224932e-11, y);
  const int q = (int)kd & 3;
  const double z = y * y;
  double r = 1.58969099521155010221e-10;
  r = fma(z, r, -2.50507602534068634195e-08);
  r = fma(z, r, 2.75573137070700676789e-06);
  r = fma(z, r, -1.98412698298579493134e-04);
  r = fma(z, r, 8.33333333332248946124e-03);
  r = fma(z, r, -1.66666666666666324348e-01);
  const double sn = fma(y * z, r, y);
  double c = -1.13596475577881948265e-11;
  c = fma(z, c, 2.08757232129817482790e-09);
  c = fma(z, c, -2.75573143513906633035e-07);
  c = fma(z, c, 2.48015872894767294178e-05);
  c = fma(z, c, -1.38888888888741095749e-03);
  c = fma(z, c, 4.16666666666666019037e-02);
  const double cs = fma(z * z, c, fma(-0.5, z, 1.0));
  dc o;
  o.im = (q & 1) ? cs : sn;
  o.re = (q & 1) ? -sn : cs;
  if (q & 2) {
    o.im = -o.im;
    o.re = -o.re;
  }
  return o;
}

// Per-phase s_memtime stamps of the demodulator (stamps build only: make stamps): thread 0 of every workgroup adds
// the ticks it spent in each phase to a global table the host prints after the launch.
#ifdef SYNC_STAMPS
__device__ unsigned long long g_demod_stamps[16];
#define DSTAMP(i)                                                        \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                \
      dst_acc[i] += now_ - dst_last;                                     \
      dst_last = now_;                                                   \
    }                                                                    \
  } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
#ifndef DEMOD_WAVES
#define DEMOD_WAVES 4  // waves per SIMD the register allocation aims at (114 registers since the preamble phasors left the symbol loop)
#endif
// Frames of four waves and more (N >= 2048) get the 128-register budget: one more workgroup per CU, spills and all
// (C5: 3.9 -> 2.95 ms with two workgroups instead of one; C3: 2.30 -> 2.05 ms with four instead of three) -- the LDS
// then has no room for the twiddle table or the carrier map, see launch_demod.  A one-wave frame (N = 512) stays at
// 168 registers: its LDS admits twelve workgroups per CU either way.
#ifndef DEMOD_WAVES_BIG
#define DEMOD_WAVES_BIG 4
#endif
__host__ __device__ constexpr int demod_waves_per_simd(int n) { return (n / 8 >= 256) ? DEMOD_WAVES_BIG : DEMOD_WAVES; }
// synchronisation of one frame's threads: a workgroup barrier -- or, where the frame is one wave sharing its workgroup
// with other frames, wave-level fences (DS instructions of a wave execute in order)
template <int FPW>
__device__ __forceinline__ void frame_sync() {
  if constexpr (FPW > 1) {
    FftWaveSync()();
  } else {
    __syncthreads();
  }
}
// TAPS = false: the kernel of a call that asked for no symbol tap (no tap pointer set, no tap pass).  The five pointers and
// the tap-pass flag are then compile-time nothing: 15 vector registers and 20 spilled scalar registers less at N = 512
// (110 -> 95 VGPRs), k_rx_demod 2.55 -> 2.46 ms at C2, 2.09 -> 2.01 at C3.
template <int N, bool TWL, bool TAPS>
__global__ void __launch_bounds__(((N / 8 < 64) ? 64 : N / 8) * demod_fpw(N), demod_waves_per_simd(N)) k_rx_demod(DemodParams q_in) {
  DemodParams q = q_in;
  if constexpr (!TAPS) {
    q.tap_sampler = nullptr;
    q.tap_fft = nullptr;
    q.tap_acq = nullptr;
    q.tap_sink = nullptr;
    q.tap_demapped = nullptr;
    q.tap_mode = 0;
  }
  static_assert(TWL || fft_onebuf(N), "up to N = 1024 the twiddles are always in LDS");
  constexpr int T = N / 8;
  constexpr int FPW = demod_fpw(N);
  extern __shared__ __align__(16) unsigned char smem[];
  const bool use_grid = q.grid != nullptr;
  // ---- tables shared by the workgroup's frames: twiddles | constellation | slicer grid | carrier map ----
  c32* twl = reinterpret_cast<c32*>(smem);  // the twiddle table (N <= 1024, longer transforms when it costs no workgroup)
  c32* cst = twl + (TWL ? fft_tw_lds_points(N) : 0);
  SlicerGrid* grid = reinterpret_cast<SlicerGrid*>(cst + q.arity);
  // the sink's carrier map, read once per carrier and symbol: in LDS, so that the symbol loop's only global loads
  // are the next symbol's samples (a wave's vector-memory counter is in order: any other load's wait would wait for
  // the prefetch too)
  int16_t* smapL = reinterpret_cast<int16_t*>(reinterpret_cast<unsigned char*>(grid) + (q.grid ? sizeof(SlicerGrid) : 0));
  // ---- this frame's own ----
  // (a frame's threads fill one wave: the slot is wave-uniform -- say so, and its LDS base lives in a scalar register)
  const int slot = FPW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T)) : 0;
  unsigned char* mine = smem + demod_lds_shared(N, TWL, q.smap_lds != 0, q.arity, q.nmap, use_grid) +
                        slot * demod_lds_frame(N, q.occ, q.nmap, q.nbits, q.shift);
  c32* fftbuf = reinterpret_cast<c32*>(mine);
  c32* Ysh = fftbuf;  // the FFT buffer is free again after the last pass: shifted spectrum, linear
  c32* hinv = fftbuf + fft_lds_points(N);
  // correlator scratch (occ + 2*shift + 1 floats): the equaliser's own array -- it is rebuilt from scratch right after
  // the correlation (barriers between)
  float* sd = reinterpret_cast<float*>(hinv);
  c32* dfe = hinv + demod_hinv_len(N, q.occ, q.shift);
  float* red = reinterpret_cast<float*>(dfe + q.occ);
  uint32_t* sbits = reinterpret_cast<uint32_t*>(red + demod_red_floats(N));
  const int sbw = demod_symbits_words(q.nmap, q.nbits);

  const int t = FPW > 1 ? (int)(threadIdx.x % T) : (int)threadIdx.x;
  const uint32_t f = blockIdx.x * FPW + (uint32_t)slot;
  const uint32_t q_nframes = dyn_nframes(q.dyn, q.nframes), q_j0 = dyn_j0(q.dyn, q.j0);

  // the shared tables, staged by the whole workgroup; the one barrier every frame of it takes part in
  {
    const int nthr = FPW * T, th = threadIdx.x;  // (= blockDim.x)
    for (int i = th; i < q.arity; i += nthr) cst[i] = q.constellation[i];
    if constexpr (TWL)
      for (int i = th; i < fft_tw_used(N); i += nthr) twl[lpad(i)] = q.tw[i];
    if (q.smap_lds)
      for (int i = th; i < q.nmap; i += nthr) smapL[i] = q.smap[i];
    if (use_grid)
      for (int i = th; i < (int)(sizeof(SlicerGrid) / 4); i += nthr)
        reinterpret_cast<uint32_t*>(grid)[i] = reinterpret_cast<const uint32_t*>(q.grid)[i];
  }
  __syncthreads();
  if (f >= q_nframes) return;  // (grid sized by an upper bound when the count lives on the device; a workgroup's last slots)
  if (q.tap_mode && q.invalid[f]) return;

  // sink state (uniform across the frame's threads)
  int sstate = 0;  // 0 search, 1 have_sync, 2 have_header
  float pll_phase = 0.f, pll_freq = 0.f;
  uint32_t nbits_total = 0;  // bits demapped since enter_have_sync
  uint32_t hdr = 0, hdr_bytes = 0;
  uint32_t packetlen = 0, header_ok = 0;
  bool done = false;  // sink went back to search
  uint32_t status = FR_INCOMPLETE;
  uint32_t end_frame = f;
  int coarse = 0;
  unsigned phase_count = 1;
  uint8_t* rawslot = q.raw + (uint64_t)f * RAW_SLOT;

  for (int i = t; i < sbw; i += T) sbits[i] = 0;
  for (int i = t; i < q.occ; i += T) {
    hinv[i] = mk(0.f, 0.f);
    dfe[i] = mk(1.f, 0.f);
  }
  frame_sync<FPW>();

#ifdef SYNC_STAMPS
  unsigned long long dst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dst_last = __builtin_amdgcn_s_memtime();
#endif
  const unsigned sidx0 = q.sign_idx[0], sidx1 = q.sign_idx[1], sidx2 = q.sign_idx[2], sidx3 = q.sign_idx[3];  // (scalar registers)
  uint32_t cf = f;  // frame whose symbols are being consumed
  for (;;) {
    const uint32_t j = q_j0 + cf;
    const uint64_t p = q.peaks[j];
    const uint32_t K = q.K[j];
    const uint64_t symb = q.sym_base[j];
    const uint64_t s00 = p - (uint64_t)N + 1;  // first sample of the preamble symbol

    // ---- NCO phasors (gr_frequency_modulator_fc closed form, float64) ---------------------------------
    // data symbols live in segment j:  phi[n] = Phi_j + st_j * (n - p + 1); for symbol k, sample t + m*T:
    //   n - p + 1 = k*L - N + 2 + t + m*T   =>   exp(j phi) = A * RL^k * RT^m
    const double st = q.step[j];
    dc A = dexpj(q.Phi[j] + st * (double)(2 - N + t));
    // (the two step phasors are the same in every lane: held in scalar registers, eight vector registers fewer)
    const dc RL = dc_uniform(dexpj(st * (double)q.L));
    const dc RT = dc_uniform(dexpj(st * (double)T));
    // the preamble symbol ends ON the flag: its samples before p belong to the previous segment(s).  (Its phasors are
    // formed inside the k == 0 branch below: they are needed once per frame, not as registers held over the symbol loop.)
    const bool pre_simple = (j == 0) || (q.peaks[j - 1] <= s00);

    // prefetch of the next symbol's samples
    c32 nx[8];
#pragma unroll
    for (int m = 0; m < 8; m++) nx[m] = q.y[s00 + (uint64_t)(t + m * T)];

    DSTAMP(0);  // frame prologue: flag data, three phasors, first prefetch issued
    dc base = A;  // A * RL^k
    for (uint32_t k = 0; k <= K; k++) {
      if (done && !q.tap_mode) break;
      // Opaque copy of the thread index: without it the compiler hoists every tid-dependent LDS / twiddle
      // address of the three FFT passes out of the symbol loop and then spills them (170+ VGPRs).
      int tl = t;
      asm volatile("" : "+v"(tl));
      c32 e[8];
#pragma unroll
      for (int m = 0; m < 8; m++) e[m] = nx[m];
      DSTAMP(6);  // (loop bookkeeping; the wait for the prefetched samples lands in the next phase)
      // ---- sigmix: chan_filt * exp(j phi[n]) ---------------------------------------------
      if (k > 0) {
        base = dmul(base, RL);
        dc r = base;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          e[m] = cmul(e[m], mk((float)r.re, (float)r.im));
          r = dmul(r, RT);
        }
      } else if (pre_simple) {
        dc Ap = {1.0, 0.0}, RTp = {1.0, 0.0};
        const dc Rflag = dexpj(q.Phi[j] + st);  // the flagged sample itself already runs on the new frequency
        if (j > 0) {
          const double stq = q.step[j - 1];
          Ap = dexpj(q.Phi[j - 1] + stq * (double)((int64_t)(s00 + (uint64_t)tl) - (int64_t)q.peaks[j - 1] + 1));
          RTp = dexpj(stq * (double)T);
        } else if (q.ref_on) {
          // chunked streams: what precedes this call's first flag runs on the NCO line carried in
          Ap = dexpj(q.ref_phi + q.ref_step * (double)((int64_t)(s00 + (uint64_t)tl) - q.ref_peak + 1));
          RTp = dexpj(q.ref_step * (double)T);
        }
        dc r = Ap;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          c32 rot = mk((float)r.re, (float)r.im);
          if (tl + m * T == N - 1) rot = mk((float)Rflag.re, (float)Rflag.im);
          e[m] = cmul(e[m], rot);
          r = dmul(r, RTp);
        }
      } else {
        // two flags closer than one FFT length: look every sample's segment up
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const uint64_t n = s00 + (uint64_t)(tl + m * T);  // (tl: not hoisted out of the symbol loop and spilled)
          int64_t i = (int64_t)j;
          while (i >= 0 && q.peaks[i] > n) i--;
          double ph = 0.0;
          if (i >= 0)
            ph = q.Phi[i] + q.step[i] * (double)(n - q.peaks[i] + 1);
          else if (q.ref_on)
            ph = q.ref_phi + q.ref_step * (double)((int64_t)n - q.ref_peak + 1);
          const dc r = dexpj(ph);
          e[m] = cmul(e[m], mk((float)r.re, (float)r.im));
        }
      }
      DSTAMP(1);  // derotation (+ the wait for this symbol's samples)
      if (q.tap_sampler) {  // ofdm_receiver-sampler_c.dat: the sampled, derotated symbol
#pragma unroll
        for (int m = 0; m < 8; m++) q.tap_sampler[(symb + k) * (uint64_t)N + (uint64_t)(tl + m * T)] = e[m];
      }
      // ---- fft_vcc(N, True, [1]*N, True): forward DFT, DC to the middle -------------------
      if constexpr (!fft_onebuf(N)) {
        if constexpr (FPW > 1)
          fft_run1<N, false, DEMOD_PK, FftWaveSync, FftTwLds>(e, tl, fftbuf, FftTwLds{twl}, FftWaveSync());
        else
          fft_run1<N, false, DEMOD_PK, FftBlockSync, FftTwLds>(e, tl, fftbuf, FftTwLds{twl}, FftBlockSync());
      } else if constexpr (TWL) {
        fft_run_tw<N, false, FftBlockSync, DEMOD_PK, FftTwLds>(e, tl, fftbuf, FftTwLds{twl}, FftBlockSync());
      } else {
        fft_run<N, false, FftBlockSync, DEMOD_PK>(e, tl, fftbuf, q.tw, FftBlockSync());
      }
      DSTAMP(2);  // transform
      // the next symbol's samples, in flight during the acquisition / sink half of this one.  (Issued after the
      // transform: its twiddle loads wait on the in-order vector-memory counter, i.e. on everything issued before them.)
      if (k < K) {
        const uint64_t s1 = s00 + (uint64_t)(k + 1) * (uint64_t)q.L;
#pragma unroll
        for (int m = 0; m < 8; m++) nx[m] = q.y[s1 + (uint64_t)(tl + m * T)];
      }
      frame_sync<FPW>();  // every thread is done reading the FFT buffers before Ysh (= buffer A) is overwritten
#pragma unroll
      for (int m = 0; m < 8; m++) Ysh[(tl + m * T + N / 2) & (N - 1)] = e[m];
      frame_sync<FPW>();
      if (q.tap_fft) {
#pragma unroll
        for (int m = 0; m < 8; m++) q.tap_fft[(symb + k) * (uint64_t)N + (uint64_t)(tl + m * T)] = Ysh[tl + m * T];
      }

      DSTAMP(3);  // next prefetch issued, shifted spectrum to LDS
      // ---- digital_ofdm_frame_acquisition ------------------------------------------------------
      if (k == 0) {
        phase_count = 1;
        // correlate(): sd[i] = |Y[i] - Y[i+2]|^2, kept only over the bins the search below reads:
        // sdl[r] = sd[zl - shift + r], r < occ + 2*shift  (0 outside [0, N-2), as the reference's zero padding)
        const int sbase = q.zl - q.shift, slen = q.occ + 2 * q.shift;
        for (int r = t; r < slen; r += T) {
          const int i = sbase + r;
          float v = 0.f;
          if (i >= 0 && i < N - 2) v = cnorm(csub(Ysh[i], Ysh[i + 2]));
          sd[r] = v;
        }
        frame_sync<FPW>();
        int index = 0;
        float mx = 0.f;
        for (int i0 = q.zl - q.shift; i0 < q.zl + q.shift; i0 += 2) {
          float pa = 0.f, pb = 0.f;
          for (int jj = t; jj < q.occ; jj += T) {
            const float kdv = q.kd[jj];
            const int ra = i0 - sbase + jj;  // in [0, slen - 1)
            pa = pa + kdv * sd[ra];
            pb = pb + kdv * sd[ra + 1];
          }
          block_sum2_f<T>(pa, pb, red);
          if (pa > mx) {
            mx = pa;
            index = i0;
          }
          if (i0 + 1 < q.zl + q.shift && pb > mx) {
            mx = pb;
            index = i0 + 1;
          }
        }
        coarse = __builtin_amdgcn_readfirstlane(index - q.zl);  // (the same in every lane, like the PLL state below)
        // calculate_equalizer()
        {
          const double a = -6.283185307179586476925 * (double)coarse * (double)q.CP / (double)N * 1.0;
          const float af = (float)a;
          c32 comp;
          det_sincosf(af, &comp.im, &comp.re);
          for (int i = 2 * t; i < q.occ; i += 2 * T) {
            const int yi = i + q.zl + coarse;
            const c32 Y = (yi >= 0 && yi < N) ? Ysh[yi] : mk(0.f, 0.f);
            hinv[i] = cdiv(q.ks[i], cmul(comp, Y));
          }
          frame_sync<FPW>();
          for (int i = 2 * t + 1; i < q.occ; i += 2 * T) {
            if (i + 1 < q.occ) {
              const c32 a1 = hinv[i + 1], a0 = hinv[i - 1];
              hinv[i] = mk((a1.re + a0.re) / 2.0f, (a1.im + a0.im) / 2.0f);
            }
          }
          frame_sync<FPW>();
          if (t == 0 && !(q.occ & 1)) hinv[q.occ - 1] = hinv[q.occ - 2];
          frame_sync<FPW>();
        }
      }
      c32 comp;
      {
        // coarse == 0 (the common case): a = -0.0, for which det_sincosf gives (+0, 1) -- skip the evaluation
        if (coarse != 0) {
          const double a = -6.283185307179586476925 * (double)coarse * (double)q.CP / (double)N * (double)phase_count;
          const float af = (float)a;
          float sn, cs;
          det_sincosf(af, &sn, &cs);
          comp = mk(f32_uniform(cs), f32_uniform(sn));
        } else {
          comp = mk(1.0f, 0.0f);
        }
        phase_count++;
        if (phase_count == 1000) phase_count = 1;
      }
      if (q.tap_acq) {
        for (int i = t; i < q.occ; i += T) {
          const int yi = i + q.zl + coarse;
          const c32 Y = (yi >= 0 && yi < N) ? Ysh[yi] : mk(0.f, 0.f);
          q.tap_acq[(symb + k) * (uint64_t)q.occ + (uint64_t)i] = cmul(cmul(hinv[i], comp), Y);
        }
      }

      DSTAMP(4);  // frame acquisition (correlation + equaliser on the preamble symbol; the phase term otherwise)
      // ---- digital_ofdm_frame_sink::work ------------------------------------------------------------
      // The symbol may end right here (preamble symbol; tap pass behind the end of the packet).  The next symbol's
      // transform then overwrites Ysh: in a frame of several waves a wave that is done with its share of the FFT /
      // acquisition taps must not get there while another still reads.  (Soak K3, seed 81: one acq-tap mismatch in
      // 16 162 cases, at N = 4096, which a replay of the same sequence did not show -- a race, found by reading for it.
      // `done` and `sstate` are the same in every thread of the frame.)
      if (done) {  // tap pass: the sink is searching again, nothing more to demap in this chain
        frame_sync<FPW>();
        continue;
      }
      if (sstate == 0) {
        // only reachable on the chain's first symbol (a preamble): enter_have_sync
        sstate = 1;
        frame_sync<FPW>();
        continue;
      }
      // demapper
      c32 carrier;
      det_sincosf(pll_phase, &carrier.im, &carrier.re);  // gr_expj(d_phase), bit-reproducible form
      carrier = mk(f32_uniform(carrier.re), f32_uniform(carrier.im));  // (the same in every lane: scalar registers)
      float are = 0.f, aim = 0.f;
      const uint32_t carry_bits = nbits_total & 7u;  // bits of the unfinished byte carried in sbits[0]
      for (int c = t; c < q.nmap; c += T) {
        // (two typed loads in two branches, not one load through a selected pointer: that would be a FLAT load, whose
        //  wait also waits for the next symbol's samples in flight)
        int i;
        if (q.smap_lds)
          i = ((const __attribute__((address_space(3))) int16_t*)smapL)[c];
        else
          i = ((const __attribute__((address_space(1))) int16_t*)q.smap)[c];
        const int yi = i + q.zl + coarse;
        const c32 Y = (yi >= 0 && yi < N) ? Ysh[yi] : mk(0.f, 0.f);
        const c32 in = cmul(cmul(hinv[i], comp), Y);
        const c32 sigrot = cmul(cmul(in, carrier), dfe[c]);
        // slicer: first minimum of |x - pos[j]|^2 over the table (digital_ofdm_frame_sink::slicer).  On a grid
        // constellation the minimum is one of the four points whose levels bracket the two parts: those four are
        // evaluated with the SAME float32 expression and the first (lowest-index) minimum among them is taken --
        // the full search's answer, at 4 distance evaluations instead of `arity`.
        unsigned best = 0;
        float bestd;
        const float sre = fabsf(sigrot.re), sim = fabsf(sigrot.im);
        if (q.sign_kind != 0 && sre > q.sign_eps && sre < q.sign_bound && sim < q.sign_bound &&
            (q.sign_kind == 1 || sim > q.sign_eps)) {
          // (selects between the four scalar entries: indexing the array with a per-lane value would make the
          //  compiler fetch it from the kernel-argument segment -- a global load, and its wait, per carrier)
          const bool rp = sigrot.re > 0.0f, ip = sigrot.im > 0.0f;
          if (q.sign_kind == 1)
            best = rp ? sidx1 : sidx0;
          else
            best = rp ? (ip ? sidx3 : sidx2) : (ip ? sidx1 : sidx0);
        } else if (use_grid && sre <= grid->bound && sim <= grid->bound) {
          int ka = 0, kb = 0;
          for (int a = 1; a < grid->nr - 1; a++) ka += (sigrot.re >= grid->lr[a]) ? 1 : 0;
          for (int b = 1; b < grid->ni - 1; b++) kb += (sigrot.im >= grid->li[b]) ? 1 : 0;
          best = grid->idx[ka * grid->ni + kb];
          bestd = cnorm(csub(sigrot, cst[best]));
#pragma unroll
          for (int c4 = 1; c4 < 4; c4++) {
            const unsigned cand = grid->idx[(ka + (c4 >> 1)) * grid->ni + kb + (c4 & 1)];
            const float dd = cnorm(csub(sigrot, cst[cand]));
            if (dd < bestd || (dd == bestd && cand < best)) {
              bestd = dd;
              best = cand;
            }
          }
        } else {
          bestd = cnorm(csub(sigrot, cst[0]));
          for (int jj = 1; jj < q.arity; jj++) {
            const float dd = cnorm(csub(sigrot, cst[jj]));
            if (dd < bestd) {
              bestd = dd;
              best = (unsigned)jj;
            }
          }
        }
        const c32 closest = cst[best];
        const c32 er = cmul_conj(sigrot, closest);
        are = are + er.re;
        aim = aim + er.im;
        const float sden = cnorm(sigrot);
        if (sden > 0.001f) {
          // closest / sigrot as conj-product times ONE reciprocal (normative: the oracle does the same)
          const float sinv = 1.0f / sden;
          const c32 qq = mk((closest.re * sigrot.re + closest.im * sigrot.im) * sinv,
                            (closest.im * sigrot.re - closest.re * sigrot.im) * sinv);
          c32 d = dfe[c];
          d.re = d.re + q.eq_gain * (qq.re - d.re);
          d.im = d.im + q.eq_gain * (qq.im - d.im);
          dfe[c] = d;
        }
        if (q.tap_sink) q.tap_sink[(symb + k) * (uint64_t)q.occ + (uint64_t)c] = sigrot;
        // LSB-first bit packing into this symbol's bit buffer
        const uint32_t bp = carry_bits + (uint32_t)c * (uint32_t)q.nbits;
        atomicOr(&sbits[bp >> 5], best << (bp & 31));
        if ((bp & 31) + (uint32_t)q.nbits > 32u) atomicOr(&sbits[(bp >> 5) + 1], best >> (32 - (bp & 31)));
      }
      DSTAMP(5);  // demapper: equalise, rotate, slice, DFE update, bit packing
      if (q.tap_demapped && t == 0) q.tap_demapped[symb + k] = 1;
      block_sum2_f<T>(are, aim, red);
      const float angle = det_atan2f(aim, are);  // arg(accumulated error), bit-reproducible form
      pll_freq = pll_freq - q.freq_gain * angle;
      pll_phase = pll_phase + pll_freq - q.phase_gain * angle;
      if (pll_phase >= 6.28318530717958647692f) pll_phase -= 6.28318530717958647692f;
      if (pll_phase < 0.0f) pll_phase += 6.28318530717958647692f;
      pll_phase = f32_uniform(pll_phase);
      pll_freq = f32_uniform(pll_freq);
      frame_sync<FPW>();  // sbits complete for this symbol
      // ---- bytes of this symbol: header parse, message bytes to the raw slot -----------------------
      const uint32_t byte0 = nbits_total >> 3;                           // index of the byte at sbits bit 0
      nbits_total += (uint32_t)q.nmap * (uint32_t)q.nbits;
      const uint32_t nbytes = nbits_total >> 3;                          // complete bytes so far
      const uint32_t nnew = nbytes - byte0;                              // complete bytes in the buffer
      const uint8_t* sb8 = reinterpret_cast<const uint8_t*>(sbits);
      // header bytes (the first four of the frame) are assembled MSB first
      for (uint32_t b = byte0; b < nbytes && b < 4; b++) {
        hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)((hdr << 8) | sb8[b - byte0]));
        hdr_bytes++;
      }
      if (sstate == 1 && hdr_bytes == 4) {
        if (((hdr >> 16) ^ (hdr & 0xFFFF)) == 0) {
          header_ok = 1;
          packetlen = (hdr >> 16) & 0x0FFF;
          sstate = 2;
        } else {
          status = FR_BAD_HEADER;
          done = true;
          end_frame = cf;
        }
      }
      if (sstate == 2) {
        // message byte i (i >= 0) is frame byte 4+i
        const uint32_t lim = 4 + packetlen;
        for (uint32_t b = byte0 + t; b < nbytes; b += T)
          if (b >= 4 && b < lim) rawslot[b - 4] = sb8[b - byte0];
        if (nbytes >= lim) {
          status = FR_COMPLETE;
          done = true;
          end_frame = cf;
        }
      }
      frame_sync<FPW>();
      // carry the unfinished byte into the next symbol's buffer, clear the rest
      {
        const uint32_t keep = (nbits_total & 7u) ? sb8[nnew] : 0u;
        frame_sync<FPW>();
        for (int i = t; i < sbw; i += T) sbits[i] = (i == 0) ? keep : 0u;
        frame_sync<FPW>();
      }
      DSTAMP(7);  // PLL reduction + update, header parse, message bytes, carry
    }
    if (done) break;
    if (cf + 1 >= q_nframes) break;
    cf++;  // the next preamble arrives while the sink is not searching: it is consumed as data
  }
  if (!done) end_frame = q_nframes - 1;

#ifdef SYNC_STAMPS
  if (threadIdx.x == 0)
    for (int i = 0; i < 8; i++) atomicAdd(&g_demod_stamps[i], dst_acc[i]);
#endif
  if (!q.tap_mode && t == 0) {
    FrameResult r;
    r.status = status;
    r.packetlen = packetlen;
    r.end_frame = end_frame;
    r.header_ok = header_ok;
    q.res[f] = r;
  }
}

// ------------------------------------------------------------------------------------
// chain resolution: frames swallowed by an earlier, still unfinished packet are invalid
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_chain_collect(const FrameResult* __restrict__ res, uint32_t nframes_s, DynFrames dyn,
                                                        uint32_t* __restrict__ list, uint32_t cap, unsigned int* count) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nframes = dyn_nframes(dyn, nframes_s);
  if (f >= nframes) return;
  if (res[f].end_frame > f) {
    const unsigned int k = atomicAdd(count, 1u);
    if (k < cap) list[k] = f;
  }
}

// single thread: sort the (short) list of chain heads, walk it, mark swallowed frames
// `pre` (chunked streams): the first npre frames were settled by earlier calls -- pre[f] != 0 says frame f was
// swallowed by a packet that began before it (possibly before this call's first sample): it is invalid and is
// no chain head here either.
__global__ void k_chain_resolve(const FrameResult* __restrict__ res, uint32_t nframes_s, DynFrames dyn, uint32_t* __restrict__ list,
                                uint32_t cap, const unsigned int* __restrict__ count, uint8_t* __restrict__ invalid,
                                unsigned int* overflow, const uint8_t* __restrict__ pre, uint32_t npre) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t nframes = dyn_nframes(dyn, nframes_s);
  for (uint32_t f = 0; f < npre && f < nframes; f++)
    if (pre[f]) invalid[f] = 1;
  unsigned int n = *count;
  if (n > cap) {
    // more chain heads than the list holds (a detector set to fire on noise): walk every frame in order instead
    int64_t cover = -1;
    for (uint32_t f = 0; f < nframes; f++) {
      if ((int64_t)f <= cover) continue;
      if (f < npre && pre[f]) continue;
      const uint32_t e = res[f].end_frame;
      if (e > f) {
        for (uint32_t g = f + 1; g <= e && g < nframes; g++) invalid[g] = 1;
        cover = (int64_t)e;
      }
    }
    return;
  }
  for (unsigned int i = 1; i < n; i++) {  // insertion sort
    const uint32_t v = list[i];
    int k = (int)i - 1;
    while (k >= 0 && list[k] > v) {
      list[k + 1] = list[k];
      k--;
    }
    list[k + 1] = v;
  }
  int64_t cover = -1;
  for (unsigned int i = 0; i < n; i++) {
    const uint32_t f = list[i];
    if ((int64_t)f <= cover) continue;  // itself swallowed: its optimistic result does not count
    if (f < npre && pre[f]) continue;   // swallowed by a packet of an earlier chunk
    const uint32_t e = res[f].end_frame;
    for (uint32_t g = f + 1; g <= e && g < nframes; g++) invalid[g] = 1;
    cover = (int64_t)e;
  }
}

// ------------------------------------------------------------------------------------
// unmake_packet
// ------------------------------------------------------------------------------------
struct DeframeParams {
  DynFrames dyn;
  uint32_t nframes;
  const FrameResult* res;
  const uint8_t* invalid;
  const uint8_t* raw;
  const uint8_t* mask;
  const uint32_t* crc_table;
  const uint32_t* xp8;  // [4097] x^(8k) mod P, reflected (crc32_combine operator for k following bytes)
  uint64_t* key;        // [nframes] (is_message << 40) | payload_bytes
  const uint64_t* pos;  // exclusive scan of key
  uint8_t* payload_out;
  uint64_t payload_cap;
  uint64_t* out_off;  // [max_pkts+1]
  uint32_t* out_len;  // [max_pkts]
  uint8_t* out_ok;    // [max_pkts]
  uint64_t* out_pos;  // [max_pkts] flag sample of the packet's preamble
  const uint64_t* peaks;  // flags of the stream; frame f belongs to peaks[j0 + f]
  uint32_t j0;
  uint32_t max_pkts;
  uint8_t* raw_tap;       // optional: concatenated messages before dewhitening
  uint64_t* counters;     // [0] headers_ok [1] packets [2] crc_ok [3] chained [4] capacity overflow
};

__global__ void __launch_bounds__(256) k_deframe_count(DeframeParams q) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= dyn_nframes(q.dyn, q.nframes)) {
    if (q.dyn.lo && f < q.dyn.npeaks) q.key[f] = 0;  // (the scan behind this kernel runs over the upper bound)
    return;
  }
  uint64_t key = 0;
  if (q.invalid[f]) {
    atomicAdd((unsigned long long*)&q.counters[3], 1ull);
  } else {
    const FrameResult r = q.res[f];
    if (r.header_ok) atomicAdd((unsigned long long*)&q.counters[0], 1ull);
    if (r.status == FR_COMPLETE) {
      const uint64_t plen = r.packetlen >= 4 ? r.packetlen - 4 : 0;
      key = (1ull << 40) | plen;
    }
  }
  q.key[f] = key;
}

// One WAVE per frame: coalesced 16-byte loads of the raw message, dewhitening, CRC-32 as 64 independent
// 16-byte CRCs per KiB combined with crc(A||B) = crc(A) * x^(8|B|) + crc(B)  (mod P), and dword-aligned
// coalesced stores of the payload through an LDS staging line.
__global__ void __launch_bounds__(256) k_deframe_write(DeframeParams q) {
  __shared__ uint32_t tab[256];
  __shared__ __align__(16) uint32_t stage_all[4][260];
  tab[threadIdx.x] = q.crc_table[threadIdx.x];
  __syncthreads();
  const int lane = lane_id(), w = wave_id();
  uint32_t* stage = stage_all[w];
  const uint32_t f = blockIdx.x * 4 + (uint32_t)w;
  if (f >= dyn_nframes(q.dyn, q.nframes)) return;
  if (q.invalid[f]) return;
  const FrameResult r = q.res[f];
  if (r.status != FR_COMPLETE) return;
  const uint64_t pos = q.pos[f];
  const uint64_t ord = pos >> 40, boff = pos & ((1ull << 40) - 1);
  const uint32_t len = r.packetlen;
  const uint32_t plen = len >= 4 ? len - 4 : 0;
  if (ord >= q.max_pkts || boff + plen > q.payload_cap) {
    if (lane == 0) atomicAdd((unsigned long long*)&q.counters[4], 1ull);
    return;
  }
  const uint8_t* msg = q.raw + (uint64_t)f * RAW_SLOT;
  const uint4* msg4 = reinterpret_cast<const uint4*>(msg);
  const uint4* mask4 = reinterpret_cast<const uint4*>(q.mask);
  uint8_t* out = q.payload_out + boff;
  uint32_t acc = 0;
  // dewhiten with offset 0 (ofdm.py:303 passes no offset) and check the CRC (crc.check_crc32)
  for (uint32_t c0 = 0; c0 < plen; c0 += 1024) {
    const uint32_t o = c0 + 16u * (uint32_t)lane;
    uint4 d = make_uint4(0, 0, 0, 0);
    if (o < plen) {
      const uint4 m = msg4[o >> 4], k = mask4[o >> 4];
      d = make_uint4(m.x ^ k.x, m.y ^ k.y, m.z ^ k.z, m.w ^ k.w);
      const uint32_t nb = (plen - o < 16u) ? (plen - o) : 16u;
      uint32_t crc = 0xFFFFFFFFu;
      const uint32_t wds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int b = 0; b < 16; b++) {
        if ((uint32_t)b < nb) {
          const uint32_t byte = (wds[b >> 2] >> (8 * (b & 3))) & 0xFFu;
          crc = tab[(crc ^ byte) & 0xFF] ^ (crc >> 8);
        }
      }
      crc ^= 0xFFFFFFFFu;
      acc ^= crc_multmodp(q.xp8[plen - (o + nb)], crc);
    }
    // ---- payload bytes of this KiB to the output, dword-aligned ------------------------------
    reinterpret_cast<uint4*>(stage)[lane] = d;
    if (lane == 0) stage[256] = 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the staging line is written
    const uint32_t rem = (plen - c0 < 1024u) ? (plen - c0) : 1024u;
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
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) acc ^= __shfl_xor(acc, d, WAVE);
  if (lane == 0) {
    int ok = 0;
    if (len >= 4) {
      const uint32_t got = ((uint32_t)(msg[plen] ^ q.mask[plen]) << 24) | ((uint32_t)(msg[plen + 1] ^ q.mask[plen + 1]) << 16) |
                           ((uint32_t)(msg[plen + 2] ^ q.mask[plen + 2]) << 8) | (uint32_t)(msg[plen + 3] ^ q.mask[plen + 3]);
      ok = (acc == got);  // crc32 of an empty payload is 0 = the empty XOR
    }
    q.out_off[ord] = boff;
    q.out_len[ord] = plen;
    q.out_ok[ord] = (uint8_t)ok;  // (packet / CRC totals are summed by the host from these flags)
    q.out_pos[ord] = q.peaks[dyn_j0(q.dyn, q.j0) + f];
  }
}

// raw (pre-dewhitening) messages, concatenated in stream order, for the PACKETS tap
__global__ void __launch_bounds__(256) k_raw_tap(DeframeParams q, const uint64_t* __restrict__ rawpos, uint8_t* __restrict__ dst) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= q.nframes) return;
  if (q.invalid[f]) return;
  const FrameResult r = q.res[f];
  if (r.status != FR_COMPLETE) return;
  const uint8_t* msg = q.raw + (uint64_t)f * RAW_SLOT;
  for (uint32_t i = 0; i < r.packetlen; i++) dst[rawpos[f] + i] = msg[i];
}
__global__ void __launch_bounds__(256) k_raw_len(DeframeParams q, uint64_t* __restrict__ lens) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= q.nframes) return;
  uint64_t l = 0;
  if (!q.invalid[f] && q.res[f].status == FR_COMPLETE) l = q.res[f].packetlen;
  lens[f] = l;
}

// Phases of the flags from their scanned integer advances.  Chunked streams: the line carried in from the
// flag that precedes this call's first one (phase ref_u at sample ref, step step_ref) is in force up to
// that first flag, which therefore starts from ref_u + turns(step_ref * (flag0 - ref)).
__global__ void __launch_bounds__(256) k_nco_phase(const uint64_t* __restrict__ peaks, const uint64_t* __restrict__ acc,
                                                    uint64_t npeaks, int ref_on, int64_t ref, uint64_t ref_u, double step_ref,
                                                    uint64_t* __restrict__ Phi_u, double* __restrict__ Phi) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npeaks) return;
  uint64_t off = 0;
  if (ref_on) off = ref_u + nco_turns(step_ref * (double)((int64_t)peaks[0] - ref));
  const uint64_t u = acc[i] + off;
  Phi_u[i] = u;
  Phi[i] = nco_radians(u);
}

// ofdm_receiver-sigmix_c.dat / -nco_c.dat (ofdm_receiver.py~:150-152): the NCO's closed form sample by sample over
// the whole stream, phi[n] = Phi_j + step_j (n - p_j + 1) for p_j <= n < p_{j+1} (0 before the first flag, or the
// line carried in / the constant of SYNC "fixed"), nco = expj(phi) rounded to float32, sigmix = chan_filt * nco.
__global__ void __launch_bounds__(256) k_sigmix_tap(const c32* __restrict__ y, uint64_t n, const uint64_t* __restrict__ peaks,
                                                     const double* __restrict__ Phi, const double* __restrict__ step,
                                                     uint64_t npeaks, int ref_on, int64_t ref_peak, double ref_phi,
                                                     double ref_step, c32* __restrict__ sigmix, c32* __restrict__ nco) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t lo = 0, hi = npeaks;  // first flag > i
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (peaks[mid] <= i) lo = mid + 1;
    else hi = mid;
  }
  double ph = 0.0;
  if (lo > 0) ph = Phi[lo - 1] + step[lo - 1] * (double)(i - peaks[lo - 1] + 1);
  else if (ref_on) ph = ref_phi + ref_step * (double)((int64_t)i - ref_peak + 1);
  const dc r = dexpj(ph);
  const c32 rot = mk((float)r.re, (float)r.im);
  if (nco) nco[i] = rot;
  if (sigmix) sigmix[i] = cmul(y[i], rot);
}

// ------------------------------------------------------------------------------------
// receive-side workspaces
// ------------------------------------------------------------------------------------
struct RxState {
  DevBuf recs, x_stage, y, metric, presel, tile_B, tile_np, tile_first, tile_pieces, avg_in, cand_u, cand_P, counters, counts, offsets,
      partial, peaks, peak_P, angle, step, inc, Phi, K, nsym, sym_base, res, raw, invalid, chain_list, key, pos,
      out_payload, out_off, out_len, out_ok, out_pos, inc_acc, Phi_u, peaks2, peak_P2, fstep, pre_inv, stash_peaks, stash_P, tap_fft, tap_acq, tap_sink, tap_demapped, raw_tap, raw_lens, raw_pos, tap_sampler, tap_sigmix, tap_nco;
  uint64_t nsamples = 0, npeaks = 0, nframes = 0, j0 = 0, nsym_total = 0, raw_tap_bytes = 0;
  const c32* y_ptr = nullptr;  // chan_filt's output of the last call: rx.y, or the input itself (SYNC "fixed")
  // ofdm_rx_submit: the input stage of the next ofdm_rx call is already queued for this buffer
  bool sub_valid = false, in_event_at_end = false;
  bool front_done = false;  // the fused front end (filter + pre-selection) of the pending call has been queued
  bool sub_hold = false;  // a submitted input stage whose buffer stays in use until the end of the ofdm_rx that picks it up
  const void* sub_iq = nullptr;
  uint64_t sub_n = 0;
  const c32* sub_dx = nullptr;
  uint64_t origin = 0;  // index, in its capture, of the first sample of the ofdm_rx calls (ofdm_rx_set_origin)
  std::vector<uint64_t> last_pos;  // host copy: flag sample of every packet of the last call
  // chunked streams (ofdm_rx_set_flag_history): flags settled by earlier calls replace whatever this call
  // detects up to trust_after; the NCO line of the flag before them
  bool nco_ref_on = false;
  int64_t nco_ref_peak = 0, nco_trust_after = 0;
  uint64_t nco_ref_u = 0;
  double nco_ref_step = 0.0;
  std::vector<uint64_t> hist_flags;
  std::vector<double> hist_steps;
  std::vector<uint8_t> hist_swallowed;
  std::vector<uint8_t> last_swallowed;  // per flag of the last call: its frame was swallowed by an earlier packet
  void release() {
    DevBuf* all[] = {&recs, &x_stage, &y,      &metric,  &presel, &tile_B,   &tile_np,  &tile_first, &tile_pieces, &avg_in,     &cand_u,
                     &cand_P,  &counters, &counts, &offsets,  &partial,  &peaks,       &peak_P,     &angle,
                     &step,    &inc,    &Phi,     &K,        &nsym,     &sym_base,    &res,        &raw,
                     &invalid, &chain_list, &key, &pos,      &out_payload, &out_off,  &out_len,    &out_ok,
                     &out_pos, &inc_acc, &Phi_u, &peaks2, &peak_P2, &fstep, &pre_inv, &stash_peaks, &stash_P, &tap_fft, &tap_acq, &tap_sink, &tap_demapped, &raw_tap, &raw_lens, &raw_pos, &tap_sampler, &tap_sigmix, &tap_nco};
    for (DevBuf* b : all) b->release();
  }
};
