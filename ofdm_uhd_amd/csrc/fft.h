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
// ds_{read,write}_b64.  Up to N = 1024 two buffers alternate so each exchange costs one
// barrier.  From N = 2048 on a transform works in ONE buffer (every middle pass reads all its
// inputs, meets a second barrier, then writes in place): 37 KB instead of 74 KB at N = 4096
// doubles the workgroups a CU can hold, which is what hides the barriers of these 4-pass sizes.
#pragma once
#include "common.h"

__host__ __device__ constexpr int fft_lds_points(int n) { return n + n / 8; }
__host__ __device__ constexpr bool fft_onebuf(int n) { return n >= 2048; }
__host__ __device__ constexpr int fft_lds_bufs(int n) { return fft_onebuf(n) ? 1 : 2; }
// LDS bytes one transform needs
__host__ __device__ constexpr int fft_lds_bytes(int n) { return fft_lds_bufs(n) * fft_lds_points(n) * (int)sizeof(c32); }

__device__ __forceinline__ int lpad(int i) { return i + (i >> 3); }

// Synchronisation of the N/8 threads of one transform.  Up to N = 512 they sit inside ONE wavefront: DS
// instructions of a wave execute in issue order, so an exchange needs no s_barrier -- only the compiler must
// keep the LDS accesses on their side of the exchange (wavefront-scope fences + the scheduling barrier).
struct FftWaveSync {
  __device__ __forceinline__ void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};
struct FftBlockSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

template <bool INV>
__device__ __forceinline__ c32 mul_mi(c32 a) {  // forward: a * (-i) ; inverse: a * (+i)
  return INV ? mk(-a.im, a.re) : mk(a.im, -a.re);
}

// X[r] = sum_q v[q] * w8^(q r), w8 = exp(-/+ 2 pi i / 8); in place
template <bool INV>
__device__ __forceinline__ void dft8(c32 v[8]) {
  const float h = 0.70710678118654752440f;
  c32 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  c32 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  c32 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  c32 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  // odd branch twiddles w8^1, w8^2, w8^3
  if (INV) {
    a5 = mk((a5.re - a5.im) * h, (a5.re + a5.im) * h);
    a7 = mk((-a7.re - a7.im) * h, (a7.re - a7.im) * h);
  } else {
    a5 = mk((a5.re + a5.im) * h, (a5.im - a5.re) * h);
    a7 = mk((a7.im - a7.re) * h, (-a7.re - a7.im) * h);
  }
  a6 = mul_mi<INV>(a6);
  c32 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  c32 b1 = cadd(a1, a3), b3 = mul_mi<INV>(csub(a1, a3));
  c32 c0 = cadd(a4, a6), c2 = csub(a4, a6);
  c32 c1 = cadd(a5, a7), c3 = mul_mi<INV>(csub(a5, a7));
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(c0, c1);
  v[5] = csub(c0, c1);
  v[3] = cadd(c2, c3);
  v[7] = csub(c2, c3);
}

template <bool INV>
__device__ __forceinline__ void dft4(c32& v0, c32& v1, c32& v2, c32& v3) {
  c32 s0 = cadd(v0, v2), d0 = csub(v0, v2);
  c32 s1 = cadd(v1, v3), d1 = mul_mi<INV>(csub(v1, v3));
  v0 = cadd(s0, s1);
  v2 = csub(s0, s1);
  v1 = cadd(d0, d1);
  v3 = csub(d0, d1);
}

__device__ __forceinline__ void dft2(c32& v0, c32& v1) {
  c32 s = cadd(v0, v1), d = csub(v0, v1);
  v0 = s;
  v1 = d;
}

// ---- the same butterflies on packed registers -------------------------------------------------------------
// A complex value in an even-aligned VGPR pair lets one v_pk_*_f32 do both parts.  clang finds the packed adds and
// scalings by itself but not the two shapes below -- "add the other operand rotated by -/+ 90 degrees" and the
// complex product with its explicit fma placement -- so they are spelled out with VOP3P operand selection (op_sel /
// op_sel_hi pick the half of each source per result half, neg_lo / neg_hi negate it).  Every result has the bits of
// the scalar code above (negations and operand swaps are exact): only the instruction count changes -- radix-8
// butterfly 26 instead of ~60, twiddle product 2 instead of 4.  Used where the transform is all a kernel does
// (k_chan_filter); in kernels bound by other work the asm blocks cost the scheduler more than they save
// (tools/experiments/README.md).
typedef float cv __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cv cv_of(c32 a) {
  cv r;
  r.x = a.re;
  r.y = a.im;
  return r;
}
__device__ __forceinline__ c32 c32_of(cv a) { return mk(a.x, a.y); }
// x + (-i) y = (x.re + y.im, x.im - y.re)
__device__ __forceinline__ cv pk_add_mi(cv x, cv y) {
  cv r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// x + (+i) y = (x.re - y.im, x.im + y.re)
__device__ __forceinline__ cv pk_add_pi(cv x, cv y) {
  cv r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
template <bool INV>
__device__ __forceinline__ cv pk_rot_add(cv x, cv y) {  // x + (-/+ i) y   (forward: -i, inverse: +i)
  return INV ? pk_add_pi(x, y) : pk_add_mi(x, y);
}
template <bool INV>
__device__ __forceinline__ cv pk_rot_sub(cv x, cv y) {  // x - (-/+ i) y
  return INV ? pk_add_mi(x, y) : pk_add_pi(x, y);
}
// cmul_f(a, w) for the forward transform, cmul_f(a, conj(w)) for the inverse, w from the forward table:
//   re = fma(a.re, w.re, -/+ a.im*w.im)    im = fma(a.re, +/- w.im, a.im*w.re)
template <bool INV>
__device__ __forceinline__ cv pk_cmul_tw(cv a, cv w) {
  cv t, r;
  if (INV) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  } else {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  }
  return r;
}
template <bool INV>
__device__ __forceinline__ void dft8_pk(cv v[8]) {
  const float h = 0.70710678118654752440f;
  const cv a0 = v[0] + v[4], a4 = v[0] - v[4];
  const cv a1 = v[1] + v[5], d5 = v[1] - v[5];
  const cv a2 = v[2] + v[6], a6 = v[2] - v[6];  // a6 still lacks its factor -/+ i: folded into c0, c2
  const cv a3 = v[3] + v[7], d7 = v[3] - v[7];
  // odd branch twiddles: w8^1 d5 = h (d5 -/+ i d5),  w8^3 d7 = -h (d7 +/- i d7)
  const cv a5 = pk_rot_add<INV>(d5, d5) * h;
  const cv a7 = pk_rot_sub<INV>(d7, d7) * (-h);
  const cv b0 = a0 + a2, b2 = a0 - a2;
  const cv b1 = a1 + a3, b3 = a1 - a3;          // b3 lacks -/+ i
  const cv c0 = pk_rot_add<INV>(a4, a6), c2 = pk_rot_sub<INV>(a4, a6);
  const cv c1 = a5 + a7, c3 = a5 - a7;          // c3 lacks -/+ i
  v[0] = b0 + b1;
  v[4] = b0 - b1;
  v[2] = pk_rot_add<INV>(b2, b3);
  v[6] = pk_rot_sub<INV>(b2, b3);
  v[1] = c0 + c1;
  v[5] = c0 - c1;
  v[3] = pk_rot_add<INV>(c2, c3);
  v[7] = pk_rot_sub<INV>(c2, c3);
}
template <bool INV>
__device__ __forceinline__ void dft4_pk(cv& v0, cv& v1, cv& v2, cv& v3) {
  const cv s0 = v0 + v2, d0 = v0 - v2;
  const cv s1 = v1 + v3, d1 = v1 - v3;  // d1 lacks -/+ i
  v0 = s0 + s1;
  v2 = s0 - s1;
  v1 = pk_rot_add<INV>(d0, d1);
  v3 = pk_rot_sub<INV>(d0, d1);
}

// twiddle exp(-/+ 2 pi i * idx / N) from the forward table
template <bool INV>
__device__ __forceinline__ c32 tw_get(const c32* __restrict__ tw, int idx) {
  c32 w = tw[idx];
  if (INV) w.im = -w.im;
  return w;
}

// One Stockham pass of radix R over sub-transforms of length LS (LS*R divides N).
// FROM_REG: inputs are e[m] = x[t + m*N/8]; otherwise read from `src` (padded LDS).
// TO_REG  : outputs end in e[m] = X[t + m*N/8] (only legal for the last pass);
//           otherwise written to `dst` (padded LDS).
// INPLACE: dst == src; `sync` is called between the last read and the first write (radix 8 only:
//           one butterfly per thread, so all of a thread's reads precede all of its writes).
// Twiddle sources.  FftTwTable reads the forward table (the usual case); FftTwRegs holds the thread's own
// twiddles of every radix-8 pass in registers (a thread's butterfly index is its thread index there, so they never
// change): for kernels that run transform after transform and cannot afford the load latency in each.
struct FftTwTable {
  const c32* __restrict__ tw;
  template <int N, int LS, int R, bool INV>
  __device__ __forceinline__ c32 get(int k, int q) const {
    return tw_get<INV>(tw, k * q * (N / (LS * R)));
  }
  template <int N, int LS, int R>
  __device__ __forceinline__ c32 fwd(int k, int q) const {
    return tw[k * q * (N / (LS * R))];
  }
};
// the forward table copied to LDS in the padded layout of the data (index i at i + i/8): for a kernel that runs many
// transforms per wave and has no registers to spare -- a short LDS read per twiddle instead of a global load whose
// wait (the vector-memory counter is in order) also waits for everything else the wave has in flight
struct FftTwLds {
  const c32* tw;  // LDS, fft_tw_lds_points(N) entries
  template <int N, int LS, int R, bool INV>
  __device__ __forceinline__ c32 get(int k, int q) const {
    c32 w = tw[lpad(k * q * (N / (LS * R)))];
    if (INV) w.im = -w.im;
    return w;
  }
  template <int N, int LS, int R>
  __device__ __forceinline__ c32 fwd(int k, int q) const {
    return tw[lpad(k * q * (N / (LS * R)))];
  }
};
// entries of the forward table the passes of an N-point transform read: indices 0 .. 7*(N/8 - 1)
__host__ __device__ constexpr int fft_tw_used(int n) { return 7 * (n / 8 - 1) + 1; }
__host__ __device__ constexpr int fft_tw_lds_points(int n) { return fft_tw_used(n) + fft_tw_used(n) / 8 + 1; }
template <int N>
struct FftTwRegs {
  static constexpr int LOG = (N == 64) ? 6 : (N == 128) ? 7 : (N == 256) ? 8 : (N == 512) ? 9 : (N == 1024) ? 10 : (N == 2048) ? 11 : 12;
  static constexpr int LS0 = (LOG % 3 == 1) ? 2 : (LOG % 3 == 2) ? 4 : 8;  // sub-length of the first twiddled pass
  static constexpr int NP = (N == 64) ? 1 : (N >= 1024) ? 3 : 2;            // twiddled (radix-8) passes
  c32 w[NP][7];
  __device__ __forceinline__ void load(const c32* __restrict__ tw, int t) {
    int ls = LS0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
      for (int q = 1; q < 8; q++) w[p][q - 1] = tw[(t % ls) * q * (N / (ls * 8))];
      ls *= 8;
    }
  }
  template <int NN, int LS, int R, bool INV>
  __device__ __forceinline__ c32 get(int, int q) const {
    static_assert(NN == N && R == 8, "register twiddles exist for the radix-8 passes");
    constexpr int p = (LS == LS0) ? 0 : (LS == LS0 * 8) ? 1 : 2;
    c32 v = w[p][q - 1];
    if (INV) v.im = -v.im;
    return v;
  }
  template <int NN, int LS, int R>
  __device__ __forceinline__ c32 fwd(int, int q) const {
    constexpr int p = (LS == LS0) ? 0 : (LS == LS0 * 8) ? 1 : 2;
    return w[p][q - 1];
  }
};

template <int N, int R, int LS, bool INV, bool FROM_REG, bool TO_REG, bool INPLACE = false, typename SyncFn = int,
          typename TwFn = FftTwTable, bool PK = false>
__device__ __forceinline__ void fft_pass_tw(c32 e[8], int t, const c32* src, c32* dst, const TwFn& twf, SyncFn sync = 0) {
  constexpr int T = N / 8;        // threads per transform
  constexpr int NB = 8 / R;       // butterflies per thread
  static_assert(!INPLACE || (NB == 1 && !FROM_REG && !TO_REG), "in-place passes are radix-8 middle passes");
  constexpr int STRIDE = N / R;   // input stride of one butterfly
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const int j = t + b * T;  // butterfly index in [0, N/R)
    c32 v[R];
    // LDS addresses as ONE padded base per butterfly plus compile-time offsets (they fold into the DS instructions'
    // offset fields): lpad(j + q*STRIDE) = lpad(j) + q*(STRIDE + STRIDE/8), STRIDE being a multiple of 8
    static_assert(FROM_REG || STRIDE % 8 == 0, "padded read offsets need STRIDE % 8 == 0");
    const int rbase = FROM_REG ? 0 : lpad(j);
#pragma unroll
    for (int q = 0; q < R; q++) {
      // j + q*STRIDE = t + (b + q*NB) * T
      if (FROM_REG)
        v[q] = e[b + q * NB];
      else
        v[q] = src[rbase + q * (STRIDE + STRIDE / 8)];
    }
    const int k = (LS == 1) ? 0 : (j % LS);
    if constexpr (PK) {
      cv pv[R];
#pragma unroll
      for (int q = 0; q < R; q++) pv[q] = cv_of(v[q]);
      if constexpr (LS > 1) {
#pragma unroll
        for (int q = 1; q < R; q++) pv[q] = pk_cmul_tw<INV>(pv[q], cv_of(twf.template fwd<N, LS, R>(k, q)));
      }
      if constexpr (R == 8) {
        dft8_pk<INV>(pv);
      } else if constexpr (R == 4) {
        dft4_pk<INV>(pv[0], pv[1], pv[2], pv[3]);
      } else {
        const cv sm = pv[0] + pv[1], df = pv[0] - pv[1];
        pv[0] = sm;
        pv[1] = df;
      }
#pragma unroll
      for (int q = 0; q < R; q++) v[q] = c32_of(pv[q]);
    } else {
      if constexpr (LS > 1) {
#pragma unroll
        for (int q = 1; q < R; q++) v[q] = cmul_f(v[q], twf.template get<N, LS, R, INV>(k, q));
      }
      if constexpr (R == 8) {
        dft8<INV>(v);
      } else if constexpr (R == 4) {
        dft4<INV>(v[0], v[1], v[2], v[3]);
      } else {
        dft2(v[0], v[1]);
      }
    }
    const int obase = (j - k) * R + k;  // (j / LS) * LS * R + k
    // lpad(obase + r*LS) = lpad(obase) + r*LS + (r*LS)/8: exact when LS is a multiple of 8, and for LS < 8 because the
    // low three bits of obase (k < LS, or R*j mod 8 in a leading pass) and of r*LS never carry into bit 3
    static_assert(LS % 8 == 0 || 8 % LS == 0, "padded write offsets");
    const int wbase = TO_REG ? 0 : lpad(obase);
    if constexpr (INPLACE) sync();
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (TO_REG)
        e[b + r * NB] = v[r];  // LS == N/R here, so obase + r*LS = j + r*STRIDE
      else
        dst[wbase + r * LS + ((r * LS) >> 3)] = v[r];
    }
  }
}

template <int N, int R, int LS, bool INV, bool FROM_REG, bool TO_REG, bool INPLACE = false, typename SyncFn = int, bool PK = false>
__device__ __forceinline__ void fft_pass(c32 e[8], int t, const c32* src, c32* dst, const c32* __restrict__ tw,
                                         SyncFn sync = 0) {
  fft_pass_tw<N, R, LS, INV, FROM_REG, TO_REG, INPLACE, SyncFn, FftTwTable, PK>(e, t, src, dst, FftTwTable{tw}, sync);
}

// Full transform.  e[m] holds x[t + m*N/8] on entry and X[t + m*N/8] on exit.
// `lds` points at this transform's 2*fft_lds_points(N) c32 scratch.  SYNC() must
// synchronise the N/8 threads of the transform (block barrier, or nothing but a
// compiler fence when they are one wave).
template <int N, bool INV, typename SyncFn, bool PK, typename TwFn>
__device__ __forceinline__ void fft_run_tw(c32 e[8], int t, c32* lds, const TwFn& tw, SyncFn sync) {
  c32* A = lds;
  c32* B = lds + fft_lds_points(N);
  if constexpr (N == 64) {
    fft_pass_tw<64, 8, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<64, 8, 8, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 128) {
    fft_pass_tw<128, 2, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<128, 8, 2, INV, false, false, false, SyncFn, TwFn, PK>(e, t, A, B, tw, sync);
    sync();
    fft_pass_tw<128, 8, 16, INV, false, true, false, SyncFn, TwFn, PK>(e, t, B, nullptr, tw, sync);
  } else if constexpr (N == 256) {
    fft_pass_tw<256, 4, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<256, 8, 4, INV, false, false, false, SyncFn, TwFn, PK>(e, t, A, B, tw, sync);
    sync();
    fft_pass_tw<256, 8, 32, INV, false, true, false, SyncFn, TwFn, PK>(e, t, B, nullptr, tw, sync);
  } else if constexpr (N == 512) {
    fft_pass_tw<512, 8, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<512, 8, 8, INV, false, false, false, SyncFn, TwFn, PK>(e, t, A, B, tw, sync);
    sync();
    fft_pass_tw<512, 8, 64, INV, false, true, false, SyncFn, TwFn, PK>(e, t, B, nullptr, tw, sync);
  } else if constexpr (N == 1024) {
    fft_pass_tw<1024, 2, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 2, INV, false, false, false, SyncFn, TwFn, PK>(e, t, A, B, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 16, INV, false, false, false, SyncFn, TwFn, PK>(e, t, B, A, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 128, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 2048) {  // one buffer
    (void)B;
    fft_pass_tw<2048, 4, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<2048, 8, 4, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<2048, 8, 32, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<2048, 8, 256, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else {  // one buffer
    static_assert(N == 4096, "unsupported FFT length");
    (void)B;
    fft_pass_tw<4096, 8, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<4096, 8, 8, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<4096, 8, 64, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<4096, 8, 512, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  }
}

template <int N, bool INV, typename SyncFn, bool PK = false>
__device__ __forceinline__ void fft_run(c32 e[8], int t, c32* lds, const c32* __restrict__ tw, SyncFn sync) {
  fft_run_tw<N, INV, SyncFn, PK, FftTwTable>(e, t, lds, FftTwTable{tw}, sync);
}

// The same transform (same butterflies, same twiddles, same results bit for bit) in ONE LDS buffer of
// fft_lds_points(N) points for every N: middle passes run in place.  sync() is also called on entry, so that
// back-to-back transforms in the same scratch are safe when the N/8 threads span more than one wave.
template <int N, bool INV, bool PK = false, typename SyncFn = int, typename TwFn = FftTwTable>
__device__ __forceinline__ void fft_run1(c32 e[8], int t, c32* A, const TwFn& tw, SyncFn sync) {
  sync();
  if constexpr (N == 64) {
    fft_pass_tw<64, 8, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<64, 8, 8, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 128) {
    fft_pass_tw<128, 2, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<128, 8, 2, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<128, 8, 16, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 256) {
    fft_pass_tw<256, 4, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<256, 8, 4, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<256, 8, 32, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 512) {
    fft_pass_tw<512, 8, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<512, 8, 8, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<512, 8, 64, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else if constexpr (N == 1024) {
    fft_pass_tw<1024, 2, 1, INV, true, false, false, SyncFn, TwFn, PK>(e, t, nullptr, A, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 2, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 16, INV, false, false, true, SyncFn, TwFn, PK>(e, t, A, A, tw, sync);
    sync();
    fft_pass_tw<1024, 8, 128, INV, false, true, false, SyncFn, TwFn, PK>(e, t, A, nullptr, tw, sync);
  } else {
    static_assert(N <= 1024, "fft_run1 is built for the channel filter's lengths; fft_run handles N >= 2048 in one buffer");
  }
}
