// fft.h -- batched small-N FFT/IFFT staged in LDS (replaces gr.fft_vcc over FFTW,
// ofdm.py:112 / ofdm_receiver.py~:126).
//
// One transform of length N is done by N/8 threads that each hold 8 points in
// registers: Stockham autosort passes of radix 8 (plus one leading radix-2 or
// radix-4 pass when N is not a power of 8).  The first pass takes its inputs
// from registers and the last leaves its outputs in registers, so a 512-point
// transform crosses LDS twice.  In both the first and the last pass thread t owns
// points  t + m*N/8, m = 0..7  -- consecutive lanes touch consecutive samples, so
// the surrounding global loads/stores are coalesced.
//
// LDS layout: point i lives at index i + i/8 (one pad per 8 points) which makes
// the strided Stockham stores and the unit-stride loads conflict-free for
// ds_{read,write}_b64.  Two buffers alternate so each exchange costs one barrier.
#pragma once
#include "common.h"

__host__ __device__ constexpr int fft_lds_points(int n) { return n + n / 8; }
// LDS bytes one transform needs (two buffers)
__host__ __device__ constexpr int fft_lds_bytes(int n) { return 2 * fft_lds_points(n) * (int)sizeof(c32); }

__device__ __forceinline__ int lpad(int i) { return i + (i >> 3); }

// ---- packed complex arithmetic ---------------------------------------------------------
// A complex value is a float2 vector in an even-aligned VGPR pair, so that one v_pk_*_f32
// does both parts.  clang emits packed adds/subs/scalings for the vector type by itself; the
// two shapes it does not find -- "add the other operand rotated by -/+90 degrees" and the
// complex product -- are spelled out with VOP3P operand selection (op_sel / op_sel_hi pick the
// half of each source per result half, neg_lo / neg_hi negate it): one instruction for a
// rotated add, two for a complex multiply (scalar code: 2 and 4).
typedef float cv __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cv cv_of(c32 a) {
  cv r;
  r.x = a.re;
  r.y = a.im;
  return r;
}
__device__ __forceinline__ c32 c32_of(cv a) { return mk(a.x, a.y); }

// x + (-i) y = (x.re + y.im, x.im - y.re)
__device__ __forceinline__ cv add_mi(cv x, cv y) {
  cv r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// x + (+i) y = (x.re - y.im, x.im + y.re)
__device__ __forceinline__ cv add_pi(cv x, cv y) {
  cv r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// forward transform: the rotation is -i; inverse: +i
template <bool INV>
__device__ __forceinline__ cv rot_add(cv x, cv y) {  // x + (-/+ i) y
  return INV ? add_pi(x, y) : add_mi(x, y);
}
template <bool INV>
__device__ __forceinline__ cv rot_sub(cv x, cv y) {  // x - (-/+ i) y
  return INV ? add_mi(x, y) : add_pi(x, y);
}
// a * w (forward) or a * conj(w) (inverse):
//   re = fma(a.re, w.re, -/+ a.im*w.im)   im = fma(a.im, w.re, +/- a.re*w.im)
template <bool INV>
__device__ __forceinline__ cv cmul_tw(cv a, cv w) {
  cv t, r;
  if (INV)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}

// X[r] = sum_q v[q] * w8^(q r), w8 = exp(-/+ 2 pi i / 8); in place.  28 packed instructions.
template <bool INV>
__device__ __forceinline__ void dft8(cv v[8]) {
  const float h = 0.70710678118654752440f;
  const cv a0 = v[0] + v[4], a4 = v[0] - v[4];
  const cv a1 = v[1] + v[5], d5 = v[1] - v[5];
  const cv a2 = v[2] + v[6], a6 = v[2] - v[6];  // a6 still lacks its factor -/+ i: folded below
  const cv a3 = v[3] + v[7], d7 = v[3] - v[7];
  // odd branch twiddles: w8^1 d5 = h (d5 -/+ i d5),  w8^3 d7 = -h (d7 +/- i d7)
  const cv a5 = rot_add<INV>(d5, d5) * h;
  const cv a7 = rot_sub<INV>(d7, d7) * (-h);
  const cv b0 = a0 + a2, b2 = a0 - a2;
  const cv b1 = a1 + a3, b3 = a1 - a3;          // b3 lacks -/+ i
  const cv c0 = rot_add<INV>(a4, a6), c2 = rot_sub<INV>(a4, a6);
  const cv c1 = a5 + a7, c3 = a5 - a7;          // c3 lacks -/+ i
  v[0] = b0 + b1;
  v[4] = b0 - b1;
  v[2] = rot_add<INV>(b2, b3);
  v[6] = rot_sub<INV>(b2, b3);
  v[1] = c0 + c1;
  v[5] = c0 - c1;
  v[3] = rot_add<INV>(c2, c3);
  v[7] = rot_sub<INV>(c2, c3);
}

template <bool INV>
__device__ __forceinline__ void dft4(cv& v0, cv& v1, cv& v2, cv& v3) {
  const cv s0 = v0 + v2, d0 = v0 - v2;
  const cv s1 = v1 + v3, d1 = v1 - v3;  // d1 lacks -/+ i
  v0 = s0 + s1;
  v2 = s0 - s1;
  v1 = rot_add<INV>(d0, d1);
  v3 = rot_sub<INV>(d0, d1);
}

__device__ __forceinline__ void dft2(cv& v0, cv& v1) {
  const cv s = v0 + v1, d = v0 - v1;
  v0 = s;
  v1 = d;
}

// One Stockham pass of radix R over sub-transforms of length LS (LS*R divides N).
// FROM_REG: inputs are e[m] = x[t + m*N/8]; otherwise read from `src` (padded LDS).
// TO_REG  : outputs end in e[m] = X[t + m*N/8] (only legal for the last pass);
//           otherwise written to `dst` (padded LDS).
template <int N, int R, int LS, bool INV, bool FROM_REG, bool TO_REG>
__device__ __forceinline__ void fft_pass(cv e[8], int t, const cv* src, cv* dst, const cv* __restrict__ tw) {
  constexpr int T = N / 8;        // threads per transform
  constexpr int NB = 8 / R;       // butterflies per thread
  constexpr int STRIDE = N / R;   // input stride of one butterfly
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const int j = t + b * T;  // butterfly index in [0, N/R)
    cv v[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
      // j + q*STRIDE = t + (b + q*NB) * T
      if (FROM_REG)
        v[q] = e[b + q * NB];
      else
        v[q] = src[lpad(j + q * STRIDE)];
    }
    const int k = (LS == 1) ? 0 : (j % LS);
    if (LS > 1) {
      constexpr int TWS = N / (LS * R);
#pragma unroll
      for (int q = 1; q < R; q++) v[q] = cmul_tw<INV>(v[q], tw[k * q * TWS]);
    }
    if (R == 8) {
      dft8<INV>(v);
    } else if (R == 4) {
      dft4<INV>(v[0], v[1], v[2], v[3]);
    } else {
      dft2(v[0], v[1]);
    }
    const int obase = (j - k) * R + k;  // (j / LS) * LS * R + k
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (TO_REG)
        e[b + r * NB] = v[r];  // LS == N/R here, so obase + r*LS = j + r*STRIDE
      else
        dst[lpad(obase + r * LS)] = v[r];
    }
  }
}

// Full transform.  e[m] holds x[t + m*N/8] on entry and X[t + m*N/8] on exit.
// `lds` points at this transform's 2*fft_lds_points(N) c32 scratch.  SYNC() must
// synchronise the N/8 threads of the transform (block barrier, or nothing but a
// compiler fence when they are one wave).
template <int N, bool INV, typename SyncFn>
__device__ __forceinline__ void fft_run_cv(cv e[8], int t, c32* lds, const c32* __restrict__ tw_, SyncFn sync) {
  cv* A = reinterpret_cast<cv*>(lds);
  cv* B = A + fft_lds_points(N);
  const cv* __restrict__ tw = reinterpret_cast<const cv*>(tw_);
  if constexpr (N == 64) {
    fft_pass<64, 8, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<64, 8, 8, INV, false, true>(e, t, A, nullptr, tw);
  } else if constexpr (N == 128) {
    fft_pass<128, 2, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<128, 8, 2, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<128, 8, 16, INV, false, true>(e, t, B, nullptr, tw);
  } else if constexpr (N == 256) {
    fft_pass<256, 4, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<256, 8, 4, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<256, 8, 32, INV, false, true>(e, t, B, nullptr, tw);
  } else if constexpr (N == 512) {
    fft_pass<512, 8, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<512, 8, 8, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<512, 8, 64, INV, false, true>(e, t, B, nullptr, tw);
  } else if constexpr (N == 1024) {
    fft_pass<1024, 2, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<1024, 8, 2, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<1024, 8, 16, INV, false, false>(e, t, B, A, tw);
    sync();
    fft_pass<1024, 8, 128, INV, false, true>(e, t, A, nullptr, tw);
  } else if constexpr (N == 2048) {
    fft_pass<2048, 4, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<2048, 8, 4, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<2048, 8, 32, INV, false, false>(e, t, B, A, tw);
    sync();
    fft_pass<2048, 8, 256, INV, false, true>(e, t, A, nullptr, tw);
  } else {
    static_assert(N == 4096, "unsupported FFT length");
    fft_pass<4096, 8, 1, INV, true, false>(e, t, nullptr, A, tw);
    sync();
    fft_pass<4096, 8, 8, INV, false, false>(e, t, A, B, tw);
    sync();
    fft_pass<4096, 8, 64, INV, false, false>(e, t, B, A, tw);
    sync();
    fft_pass<4096, 8, 512, INV, false, true>(e, t, A, nullptr, tw);
  }
}

// c32 front end: e[m] holds x[t + m*N/8] on entry and X[t + m*N/8] on exit
template <int N, bool INV, typename SyncFn>
__device__ __forceinline__ void fft_run(c32 e[8], int t, c32* lds, const c32* __restrict__ tw, SyncFn sync) {
  cv v[8];
#pragma unroll
  for (int m = 0; m < 8; m++) v[m] = cv_of(e[m]);
  fft_run_cv<N, INV>(v, t, lds, tw, sync);
#pragma unroll
  for (int m = 0; m < 8; m++) e[m] = c32_of(v[m]);
}
