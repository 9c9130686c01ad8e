// common.h -- device helpers shared by the OFDM kernels (gfx950 only).
//
// Arithmetic rule of this code base: the translation unit is compiled with
// -ffp-contract=off, so `a*b + c` is two roundings exactly like the GNU Radio
// blocks it restates; every fused multiply-add is an explicit fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ofdm_hip.h"

#define WAVE 64

struct c32 {
  float re, im;
};
static_assert(sizeof(c32) == 8, "c32 must be an interleaved float pair");

__device__ __forceinline__ c32 mk(float re, float im) {
  c32 z;
  z.re = re;
  z.im = im;
  return z;
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return mk(a.re - b.re, a.im - b.im); }
// gr_complex product: two products and one add per part, separately rounded
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
__device__ __forceinline__ c32 cmul_conj(c32 a, c32 b) {  // a * conj(b)
  return mk(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im);
}
__device__ __forceinline__ c32 cdiv(c32 a, c32 b) {
  float den = b.re * b.re + b.im * b.im;
  return mk((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
__device__ __forceinline__ float cnorm(c32 a) { return a.re * a.re + a.im * a.im; }
// fused complex multiply for the FFT butterflies (tolerance-level arithmetic)
__device__ __forceinline__ c32 cmul_f(c32 a, c32 b) {
  return mk(fmaf(a.re, b.re, -(a.im * b.im)), fmaf(a.re, b.im, a.im * b.re));
}

// a(x) * b(x) mod P(x) in the reflected representation zlib uses for crc32_combine (x^0 = bit 31):
// crc(A || B) = crc(A) * x^(8|B|) + crc(B)  (mod P) lets 64 lanes check-sum 16-byte pieces independently.
__device__ __forceinline__ uint32_t crc_multmodp(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    p ^= (0u - ((a >> (31 - i)) & 1u)) & b;
    b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
  }
  return p;
}

// ---- wave / block scans -------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_incl_scan_add(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    T o = __shfl_up(v, d, WAVE);
    if (lane >= d) v += o;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

// Exclusive block scan (sum).  `scratch` holds one T per wave (+1).  All threads
// of the block must call it; contains two __syncthreads().  Returns the exclusive
// prefix of `v`; *total receives the block total.
template <typename T>
__device__ __forceinline__ T block_excl_scan_add(T v, T* scratch, T* total) {
  const int lane = lane_id(), w = wave_id();
  const int nw = (blockDim.x + WAVE - 1) / WAVE;
  T inc = wave_incl_scan_add(v);
  if (lane == WAVE - 1) scratch[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    T s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---- Q23.40 fixed point for the order-independent moving sums -----------------
#define Q40_SCALE 1099511627776.0          /* 2^40 */
#define Q40_INV (1.0 / 1099511627776.0)    /* 2^-40 */

// llrint(v * 2^40) for |v| <= 1024, bit-identical to the CPU's llrint: adding
// 1.5*2^52 makes the FPU do the round-to-nearest-even to an integer, whose value
// then sits in the low mantissa bits.
__device__ __forceinline__ long long q40_from_float(float v) {
  double t = (double)v * Q40_SCALE + 6755399441055744.0;  // 1.5 * 2^52
  long long b = __double_as_longlong(t);
  return (b & 0x000FFFFFFFFFFFFFll) - 0x0008000000000000ll;
}
// llrint(x * 2^40) of a float64 |x| < 2048 (the scaling is exact; the addition rounds to nearest even)
__device__ __forceinline__ long long q40_from_double(double x) {
  double t = x * Q40_SCALE + 6755399441055744.0;  // 1.5 * 2^52
  long long b = __double_as_longlong(t);
  return (b & 0x000FFFFFFFFFFFFFll) - 0x0008000000000000ll;
}
__device__ __forceinline__ long long q40_clamped(float v) {
  v = fminf(fmaxf(v, -511.0f), 511.0f);
  return q40_from_float(v);
}
__device__ __forceinline__ double q40_to_double(long long s) { return (double)s * Q40_INV; }

// ---- complex_to_arg for the sample-and-held fine-frequency estimate ------------------------
// Plain float32 operations (Cephes-style atanf, ~2 ulp), NOT ocml's atan2f: the CPU restatement
// evaluates the same operations and gets the same bits.  The NCO integrates this angle over
// thousands of samples; a 1-ulp difference between two libm's would grow to 1e-4 rad.
__device__ __forceinline__ float det_atanf_pos(float x) {
  float y0;
  if (x > 2.414213562373095f) {
    y0 = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) {
    y0 = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y0 = 0.0f;
  }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = p * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  p = p * z;
  p = p * x + x;
  return y0 + p;
}
__device__ __forceinline__ float det_atan2f(float y, float x) {
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966f;
    if (y < 0.0f) return -1.5707963267948966f;
    return 0.0f;
  }
  const float a = det_atanf_pos(fabsf(y / x));
  const float r = (x > 0.0f) ? a : (3.14159265358979323846f - a);
  return (y < 0.0f) ? -r : r;
}

// ---- sin / cos, bit-reproducible (same plain IEEE operations as the CPU restatement) ---------------
// float32 argument of any size the receiver produces (|x| < ~1e6): reduction by multiples of pi/2 in float64
// (two-part pi/2), Cephes sinf / cosf polynomials on [-pi/4, pi/4]; <= 2 ulp.
__device__ __forceinline__ void det_sincosf(float x, float* sn, float* cs) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.63661977236758134308);
  double yd = fma(-kd, 1.57079632673412561417e+00, xd);
  yd = fma(-kd, 6.07710050650619224932e-11, yd);
  const float y = (float)yd;
  const int q = (int)kd & 3;
  const float z = y * y;
  float ps = -1.9515295891e-4f;
  ps = ps * z + 8.3321608736e-3f;
  ps = ps * z - 1.6666654611e-1f;
  ps = ps * z;
  ps = ps * y + y;
  float pc = 2.443315711809948e-5f;
  pc = pc * z - 1.388731625493765e-3f;
  pc = pc * z + 4.166664568298827e-2f;
  pc = pc * z;
  pc = pc * z - 0.5f * z;
  pc = pc + 1.0f;
  float s_ = (q & 1) ? pc : ps;
  float c_ = (q & 1) ? -ps : pc;
  if (q & 2) {
    s_ = -s_;
    c_ = -c_;
  }
  *sn = s_;
  *cs = c_;
}

// ---- misc -----------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t pad_symbol_hash(uint64_t seed, uint64_t pkt, uint64_t slot,
                                                             uint32_t arity) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (pkt + 1) + 0xBF58476D1CE4E5B9ull * (slot + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)((z >> 32) % arity);
}

// Synthetic channel noise (stands in for the radio pair; mirrored by the oracle).  Philox-2x32-7 (Salmon et
// al., SC'11; seven rounds is Random123's own reduced-round variant, known-answer vectors in tests/test_oracle.py)
// keyed by (seed, stream); counter = sample index / 2: one call yields two 32-bit words, word 0 for the even sample
// of the pair and word 1 for the odd one.  A sample's word gives 16 bits of Box-Muller radius and 16 bits of angle --
// plenty for a test channel at 30 dB.  3.5 integer multiplies per sample (round 1: ten rounds per sample).
#define CHAN_PHILOX_ROUNDS 7
__host__ __device__ __forceinline__ uint32_t chan_key(uint64_t seed, uint64_t stream) {
  return (uint32_t)seed ^ (uint32_t)(seed >> 32) ^ ((uint32_t)stream * 0x9E3779B9u + (uint32_t)(stream >> 32) * 0x85EBCA6Bu);
}
__device__ __forceinline__ void philox2x32(uint32_t& c0, uint32_t& c1, uint32_t k) {
#pragma unroll
  for (int r = 0; r < CHAN_PHILOX_ROUNDS; r++) {
    const uint64_t pr = (uint64_t)0xD256D193u * (uint64_t)c0;  // one v_mad_u64_u32 for both halves
    c0 = (uint32_t)(pr >> 32) ^ k ^ c1;
    c1 = (uint32_t)pr;
    k += 0x9E3779B9u;
  }
}
// both noise words of the pair that holds sample idx
__device__ __forceinline__ void chan_pair_words(uint64_t idx, uint32_t key, uint32_t& w_even, uint32_t& w_odd) {
  const uint64_t pi = idx >> 1;
  w_even = (uint32_t)pi;
  w_odd = (uint32_t)(pi >> 32);
  philox2x32(w_even, w_odd, key);
}
__device__ __forceinline__ c32 chan_add_noise(c32 x, uint32_t word, float sigma) {
  const float inv16 = 1.0f / 65536.0f;
  const float u1 = ((float)(word >> 16) + 0.5f) * inv16;
  const float u2 = ((float)(word & 0xFFFFu) + 0.5f) * inv16;
  // Box-Muller on the hardware transcendental units: v_log_f32 (log2), v_sqrt_f32 and v_sin/v_cos_f32, whose
  // argument is in revolutions -- exactly u2.  A few 1e-7 off libm, scaled by sigma: far below the float32
  // resolution of the signal it is added to.
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln(u1) = -2 ln2 log2(u1)
  const float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);
  const float s = sigma * 0.70710678118654752440f;
  x.re = x.re + s * (rad * cs);
  x.im = x.im + s * (rad * sn);
  return x;
}
__device__ __forceinline__ c32 chan_rotate(c32 x, uint64_t idx, float cfo) {
  double ph = (double)cfo * (double)idx;
  ph = ph - 6.283185307179586476925 * floor(ph / 6.283185307179586476925 + 0.5);
  double s, c;
  sincos(ph, &s, &c);
  return cmul(x, mk((float)c, (float)s));
}

// one channel use: rotate by the carrier offset and add circular Gaussian noise
__device__ __forceinline__ c32 channel_apply(c32 x, uint64_t idx, float sigma, float cfo, uint64_t seed,
                                             uint64_t stream) {
  if (cfo != 0.0f) x = chan_rotate(x, idx, cfo);
  if (sigma > 0.0f) {
    uint32_t we, wo;
    chan_pair_words(idx, chan_key(seed, stream), we, wo);
    x = chan_add_noise(x, (idx & 1) ? wo : we, sigma);
  }
  return x;
}
