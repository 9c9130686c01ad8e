// sense.h -- the spectrum sensor of predictive_sense.py (:72-123 flowgraph, :150-222
// sense_loop, :235-268 hex_conv) as two kernels.
//
// k_sense<NS>   stream_to_vector(NS) -> fft_vcc(NS, True, window) -> complex_to_mag_squared
//               -> bin_statistics_f's max-hold, fused: one workgroup runs G transforms of
//               length NS side by side (NS/8 threads each, 8 points per thread in
//               registers, fft.h), keeps the running per-bin maximum in registers over its
//               share of the dwell vectors, folds the G groups through LDS and merges
//               into the message body.  Every IQ sample of a dwell is read from HBM
//               exactly once (8 B/sample, coalesced: lane t reads sample t + m*NS/8) and
//               nothing but the NS-float message is written.
// k_sense_decide  sense_loop's tail: float64 sum of avg_msgs messages in message order,
//               /avg_msgs, threshold, half swap, LSB-first nibble -> hex character.
//
// The merge across workgroups uses atomicMax on the float bit patterns: powers are >= 0
// and never NaN (the register max ignores NaN exactly like accrue_stats' `>` test), so the
// unsigned order is the float order and the result does not depend on arrival order.
#pragma once
#include "common.h"
#include "fft.h"
#ifndef SENSE_PK
#define SENSE_PK false  // hand-packed butterflies (fft.h)
#endif

struct SenseParams {
  const c32* x;        // IQ stream
  const float* win;    // NS window taps
  const c32* tw;       // NS forward twiddles exp(-2 pi i k / NS)
  float* msgs;         // [nmsgs][NS], zero-initialised
  uint32_t tune_delay, dwell_delay;
  uint32_t nsplit;     // workgroups per message
  uint64_t msg0;       // first message of this launch (grid.y is limited to 65535)
};

constexpr int sense_threads(int ns) { return ns / 8 < 256 ? 256 : ns / 8; }
constexpr int sense_groups(int ns) { return sense_threads(ns) / (ns / 8); }
// Long transforms (a workgroup of 256 / 512 threads per vector, NS >= 2048) read their twiddles from a table in LDS and
// fetch the window taps anew every round: with both in registers the kernel needs 184 of them -- ONE workgroup per CU,
// eight waves that spend their time at the transform's barriers (C5: 2.0 ms alone, 3.9 ms beside the demodulator).
constexpr bool sense_lean(int ns) { return ns >= 2048; }
constexpr int sense_lds_bytes(int ns) {
  return sense_groups(ns) * fft_lds_bytes(ns) + (sense_lean(ns) ? fft_tw_lds_points(ns) * (int)sizeof(c32) : 0);
}

template <int NS>
__global__ void __launch_bounds__(sense_threads(NS), sense_lean(NS) ? 4 : 1) k_sense(SenseParams p) {
  constexpr int TPT = NS / 8;          // threads per transform
  constexpr int G = sense_groups(NS);  // transforms in flight per workgroup
  extern __shared__ __align__(16) unsigned char smem_raw[];
  c32* lds = reinterpret_cast<c32*>(smem_raw);
  const int tid = threadIdx.x;
  const int g = tid / TPT, t = tid % TPT;
  const uint64_t msg = p.msg0 + blockIdx.y;
  const uint64_t period = (uint64_t)p.tune_delay + p.dwell_delay;
  const uint64_t v0 = msg * period + p.tune_delay;  // first accrued vector of this message

  constexpr bool LEAN = sense_lean(NS);
  float w[8], mx[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    w[m] = p.win[t + m * TPT];
    mx[m] = 0.0f;  // reset_stats()
  }
  c32* my = lds + (size_t)g * fft_lds_bufs(NS) * fft_lds_points(NS);
  // a thread keeps its place in its transform for the whole dwell: its twiddles live in registers.  (Fetched per round
  // they would be global loads whose waits -- the vector-memory counter is in order -- also wait for the prefetch.)
  FftTwRegs<LEAN ? 64 : NS> twr;
  c32* twl = lds + (size_t)G * fft_lds_bufs(NS) * fft_lds_points(NS);  // (lean) the twiddle table
  if constexpr (LEAN) {
    for (int i = tid; i < fft_tw_used(NS); i += sense_threads(NS)) twl[lpad(i)] = p.tw[i];
    __syncthreads();
  } else {
    twr.load(p.tw, t);
  }
  const uint32_t stride = p.nsplit * G;
  // every group of the workgroup runs the same number of rounds (barriers inside fft_run)
  const uint32_t rounds = (p.dwell_delay + stride - 1) / stride;
  // software pipeline: the loads of round r+1 are in flight while round r is transformed
  c32 nx[8];
  {
    const uint32_t f = blockIdx.x * G + g;
    const bool live = f < p.dwell_delay;
    const c32* src = p.x + (v0 + (live ? f : 0)) * (uint64_t)NS;
#pragma unroll
    for (int m = 0; m < 8; m++) nx[m] = live ? src[t + m * TPT] : mk(0.f, 0.f);
  }
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t f = r * stride + blockIdx.x * G + g;
    const bool live = f < p.dwell_delay;
    c32 e[8];
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = mk(nx[m].re * w[m], nx[m].im * w[m]);  // fft_vcc: in[i] * window[i]
    {
      const uint32_t fn = f + stride;
      const bool ln = (r + 1 < rounds) && fn < p.dwell_delay;
      const c32* src = p.x + (v0 + (ln ? fn : 0)) * (uint64_t)NS;
#pragma unroll
      for (int m = 0; m < 8; m++) nx[m] = ln ? src[t + m * TPT] : mk(0.f, 0.f);
    }
    if constexpr (LEAN) {
      int tt = t;  // opaque copy, renewed every round: keeps the passes' LDS addresses out of the registers
      asm volatile("" : "+v"(tt));
      fft_run_tw<NS, false, FftBlockSync, SENSE_PK, FftTwLds>(e, tt, my, FftTwLds{twl}, FftBlockSync());
      // the window taps of the next round (L2): fetched behind the transform, where the registers are free again
#pragma unroll
      for (int m = 0; m < 8; m++) w[m] = p.win[tt + m * TPT];
    } else if constexpr (TPT <= WAVE) {
      fft_run_tw<NS, false, FftWaveSync, SENSE_PK, FftTwRegs<NS>>(e, t, my, twr, FftWaveSync());
    } else {
      fft_run_tw<NS, false, FftBlockSync, SENSE_PK, FftTwRegs<NS>>(e, t, my, twr, FftBlockSync());
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      float pw = e[m].re * e[m].re + e[m].im * e[m].im;  // complex_to_mag_squared
      mx[m] = (live && pw > mx[m]) ? pw : mx[m];          // accrue_stats
    }
    if constexpr (TPT > WAVE) __syncthreads();  // LDS scratch is reused by the next round (wave-private up to NS = 512)
  }
  // fold the G groups: bin b of group g at fl[g*NS + b]
  float* fl = reinterpret_cast<float*>(smem_raw);
  if (G > 1) {
    __syncthreads();  // every group is done with its transform scratch, which fl overlays
#pragma unroll
    for (int m = 0; m < 8; m++) fl[g * NS + t + m * TPT] = mx[m];
    __syncthreads();
    for (int b = tid; b < NS; b += sense_threads(NS)) {
      float v = fl[b];
      for (int q = 1; q < G; q++) v = fmaxf(v, fl[q * NS + b]);
      atomicMax(reinterpret_cast<unsigned int*>(p.msgs + msg * NS + b), __float_as_uint(v));
    }
  } else {
#pragma unroll
    for (int m = 0; m < 8; m++)
      atomicMax(reinterpret_cast<unsigned int*>(p.msgs + msg * NS + t + m * TPT), __float_as_uint(mx[m]));
  }
}

struct SenseDecideParams {
  const float* msgs;  // [nmsgs][S]
  double* mean;       // [ndec][S] ascending frequency
  uint8_t* bits;      // [ndec][S]
  char* hex;          // [ndec][S/4]
  uint32_t S, avg_msgs, skip_msgs;
  double threshold;
};

__global__ void __launch_bounds__(256) k_sense_decide(SenseDecideParams p) {
  __shared__ uint8_t ino[OFDM_SENSE_MAX_FFT];
  const uint64_t d = blockIdx.x;
  const uint32_t S = p.S, H = S / 2;
  const uint64_t per = (uint64_t)p.avg_msgs + p.skip_msgs;
  for (uint32_t i = threadIdx.x; i < S; i += 256) {
    double acc = 0.0;
    for (uint32_t k = 0; k < p.avg_msgs; k++) acc = acc + (double)p.msgs[(d * per + k) * S + i];
    acc = acc / (double)p.avg_msgs;
    const uint8_t bit = acc > p.threshold ? 0 : 1;
    const uint32_t o = i < H ? i + H : i - H;  // half swap: FFT order -> ascending frequency
    ino[o] = bit;
    p.mean[d * S + o] = acc;
    p.bits[d * S + o] = bit;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < S / 4; j += 256) {
    const int v = ino[4 * j] | (ino[4 * j + 1] << 1) | (ino[4 * j + 2] << 2) | (ino[4 * j + 3] << 3);
    p.hex[d * (S / 4) + j] = "0123456789ABCDEF"[v];
  }
}
