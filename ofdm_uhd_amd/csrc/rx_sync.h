// rx_sync.h -- receive-side synchronisation kernels.
//
//   k_sync      chan_filt (gr_fft_filter_ccc, ofdm_receiver.py~:76,131) + ofdm_sync_pn's
//               metric chain (ofdm_receiver.py~:97-101) in one streaming pass: each
//               workgroup walks a segment of the stream tile by tile, keeping the
//               filter / correlator history in LDS.  Emits the filtered stream y and,
//               instead of a per-sample metric, only the sparse "candidate" samples the
//               peak detector can ever act on.
//   k_avg_carry per-tile carry-in of the peak detector's running average.
//   k_peak      gr_peak_detector_fb run independently on every candidate interval.
//   scans       small device-wide exclusive scans used by the bookkeeping kernels.
//
// Why candidates are enough: the detector's average obeys avg = a*u + (1-a)*avg on EVERY
// sample regardless of state, and u = Mbar - 1 >= -1, so avg >= -1 and both thresholds
// avg*rise, avg*fall are >= -max(rise, fall) =: theta.  A run can only start, continue or
// record a new peak on samples with u > theta; any sample with u <= theta closes an open
// run.  Hence maximal intervals {u > theta} are independent sub-problems once avg at their
// first sample is known, and avg is a linear recurrence (tile summaries + look-back).
//
// Two-speed metric.  The timing flag is an arg-max of u, so u must carry the oracle's exact
// bits wherever the detector can look -- but that is ~1 % of the stream.  Every tile first
// gets a cheap float32 evaluation of the metric (moving sums re-anchored per tile, so the
// rounding error stays ~1e-6 and does not drift); samples with u_approx > theta - GUARD mark
// a neighbourhood that is then re-evaluated in the normative Q23.40 fixed point (bit-exact,
// evaluation-order independent).  The float32 values also feed the detector's running
// average, whose rounding in the reference is of the same order.
#pragma once
#include "common.h"
#include "fft.h"

#ifndef SYNC_THREADS
#define SYNC_THREADS 256
#endif
#define SYNC_V 8                            // consecutive samples per thread
#define SYNC_TILE (SYNC_THREADS * SYNC_V)   // 2048 samples per tile
#define SYNC_GUARD 1.0e-3f                  // guard band of the float32 pre-selection
#define SYNC_ILL 0.015625f                  // window energy below 1/64 of its running maximum: float32 has lost it
#define SYNC_CHUNK_C 4096                   // candidates per allocation chunk (>= SYNC_TILE)
#ifndef SYNC_MAX_WG
#define SYNC_MAX_WG 5                        // most k_sync workgroups per CU the register budgets are built for
#endif
#define SYNC_CHUNK_P 1024                   // pieces per allocation chunk (>= SYNC_TILE / 2)

struct SyncPiece {
  uint64_t start;    // absolute sample index of the first candidate of the piece
  uint64_t end;      // absolute index of its last candidate
  uint64_t val_off;  // offset of its first sample in the candidate value arrays
  double bloc;       // zero-initialised running average over the tile's samples before `start`
};

// a tile whose float32 metric came near the threshold: work item of k_sync_exact
struct SyncRec {
  uint64_t tile;
  int amin, bmax;      // range of the approximate candidates, tile-relative
  float gpre, gpost;   // float32 detector summary of the samples before amin / after bmax, weighted to the tile's end
};

struct SyncParams {
  int N, D, CP;
  int HY;         // y history the metric needs before a tile (2*D, multiple of 8)
  int HM;         // M history (CP)
  int R;          // samples in the LDS ring of y (multiple of 8)
  int tiles_per_seg;
  int tap_only;   // k_sync_exact: metric tap pass -- every tile's whole range re-evaluated into metric_tap, nothing else written
  int ablate;     // diagnostic build only (-DSYNC_DIAG, see SYNC_ABLATE): 2 skip metric
  unsigned long long* stamps;  // diagnostic build (-DSYNC_STAMPS): [wg][12] cycles per phase of wave 0
  uint64_t nsamples, ntiles;
  float tapcp;       // float(1/CP)
  float cand_thr;    // -max(rise, fall)
  float alpha;       // peak detector alpha
  double decay;      // double(1.0f - alpha)
  const c32* y;       // filtered stream (k_chan_filter's output)
  float* metric_tap;  // optional [nsamples]: the normative metric (k_sync_exact's tap pass)
  float* presel_tap;  // optional [nsamples]: the float32 pre-selection (k_sync)
  // tables of the normative average (host, create_impl): dpow[j] = decay^j by repeated multiplication, j = 0..SYNC_TILE;
  // ipow[j] = 1 / dpow[j]; wtab[t] = float(dpow[SYNC_TILE - SYNC_V (t + 1)]): weight of lane t's chain in a full tile
  const double* dpow;
  const double* ipow;
  const float* wtab;
  // outputs
  double* tile_B;          // [ntiles] zero-init running average over the tile
  uint32_t* tile_npieces;  // [ntiles] candidate pieces of the tile ...
  uint64_t* tile_first;    // [ntiles] ... stored at pieces[tile_first .. +tile_npieces)
  // candidate / piece storage is handed out in chunks (one atomic per SYNC_CHUNK_C candidates instead of
  // one per tile: no round-trip latency on the common path)
  SyncPiece* pieces;       // [piece_cap]
  uint64_t piece_cap;
  unsigned long long* piece_count;  // device counter
  float* cand_u;           // [cand_cap]
  c32* cand_P;             // [cand_cap]
  uint64_t cand_cap;
  unsigned long long* cand_count;  // device counter
  unsigned int* overflow;          // device flag
  int exact_small;                 // k_sync_exact: ranges up to this many samples go to the wave-sized workgroups
  SyncRec* recs;                   // [ntiles]
  unsigned long long* rec_count;   // device counter
};

__host__ __device__ inline int sync_lp(int i) { return i + (i >> 3); }

// Filter geometry (host and device): gr_fft_filter_ccc sizes its transform as 2 * 2^ceil(log2(ntaps)) and
// produces F - ntaps + 1 outputs per block; transforms shorter than 64 points are not built.
__host__ __device__ inline int sync_filter_F(int ntaps) {
  int p2 = 1;
  while (p2 < ntaps) p2 <<= 1;
  const int f = 2 * p2;
  return f < 64 ? 64 : f;
}
// ring of y in k_sync's LDS: the history a tile's metric looks back on and the tile itself
__host__ __device__ inline int sync_ring_samples(int HY) { return (HY + SYNC_TILE + 7) / 8 * 8; }

// ---------------------------------------------------------------------------------
// Channel filter = gr_fft_filter_ccc(1, taps) (ofdm_receiver.py~:76,131) the way GNU Radio runs it [SURVEY A.5]:
// overlap-save blocks of B = F - ntaps + 1 outputs on the grid m*B of the capture; block b transforms the F samples
// x[b*B - (ntaps-1) .. b*B + B), multiplies by the transformed taps (scaled 1/F) and keeps the last B points of
// the inverse transform.  One block = F/8 threads (one wave at F = 512), 8 points per thread in registers, the two
// exchanges of each transform through a private LDS scratch (no workgroup barrier up to F = 512); a workgroup runs
// 256/(F/8) blocks side by side and walks the stream round by round, the next round's window in flight while the
// current one is transformed.  A thread keeps its place in its block for the whole kernel, so its twiddles and its
// bins of the transformed taps live in registers.  Streaming: 8 B in + 8 B out per sample, 4.6 KB of LDS per wave.
// The transform schedule is fft.h's (the oracle runs the identical schedule: y is bit-exact).
// ---------------------------------------------------------------------------------
struct FilterParams {
  int B;       // outputs per block = F - ntaps + 1
  int ntm1;    // ntaps - 1 = F - B
  int goff;    // blocks start at samples m*B - goff (goff in [0, B)): grid of the capture's first sample (ofdm_rx_set_origin)
  uint64_t nsamples, nrounds;
  const c32* x;
  c32* y;
  const c32* Hf;   // [F] transform of the taps, scaled by 1/F
  const c32* twF;  // [F] exp(-2 pi i k / F)
};

#ifndef FILTER_W
#define FILTER_W 4
#endif
#ifndef FILTER_PK
#define FILTER_PK true  // hand-packed butterflies (fft.h): the transform is all this kernel does
#endif
template <int F>
__global__ void __launch_bounds__(256, FILTER_W) k_chan_filter(FilterParams p) {
  constexpr int TF = F / 8;     // threads per block of the filter
  constexpr int BPR = 256 / TF; // blocks per round of a workgroup
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int g = tid / TF, t = tid % TF;
  c32* sc = reinterpret_cast<c32*>(smem) + g * fft_lds_points(F);
  const int B = p.B, ntm1 = p.ntm1;
  FftTwRegs<F> twr;
  twr.load(p.twF, t);
  c32 Hr[8];
#pragma unroll
  for (int m = 0; m < 8; m++) Hr[m] = p.Hf[t + m * TF];
  // first output sample of this thread's block in round r: (r * BPR + g) * B - goff.  A round whose windows and outputs
  // all lie inside the stream (every round but the first and the last few) needs no per-sample bounds test: one
  // pointer per thread plus constant offsets.
  auto interior = [&](uint64_t rr) -> bool {
    const int64_t lo = (int64_t)(rr * BPR * (uint64_t)B) - p.goff - ntm1;
    const int64_t hi = (int64_t)((rr * BPR + BPR - 1) * (uint64_t)B) - p.goff + B;  // one past the round's last output
    return lo >= 0 && (uint64_t)hi <= p.nsamples;
  };
  auto load_window = [&](c32 (&xn)[8], uint64_t rr) {
    const int64_t x0 = (int64_t)((rr * BPR + (uint64_t)g) * (uint64_t)B) - p.goff - ntm1 + t;
    if (interior(rr)) {  // (uniform)
      const c32* px = p.x + x0;
#pragma unroll
      for (int m = 0; m < 8; m++) xn[m] = px[m * TF];
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int64_t xi = x0 + m * TF;
        xn[m] = (xi >= 0 && (uint64_t)xi < p.nsamples) ? p.x[xi] : mk(0.f, 0.f);
      }
    }
  };
  // one round: window e -> transform -> x transformed taps -> inverse -> the last B points out
  auto do_round = [&](c32 (&e)[8], uint64_t r) {
    // Opaque copy of the thread's place in its block, renewed every round: otherwise the compiler hoists the
    // LDS addresses of all passes out of the loop (~40 registers) and spills.
    int tt = t;
    asm volatile("" : "+v"(tt));
    if constexpr (TF <= WAVE) {
      fft_run1<F, false, FILTER_PK>(e, tt, sc, twr, FftWaveSync());
    } else {
      fft_run1<F, false, FILTER_PK>(e, tt, sc, twr, FftBlockSync());
    }
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = cmul(e[m], Hr[m]);  // volk_32fc_x2_multiply_32fc
    asm volatile("" : "+v"(tt));
    if constexpr (TF <= WAVE) {
      fft_run1<F, true, FILTER_PK>(e, tt, sc, twr, FftWaveSync());
    } else {
      fft_run1<F, true, FILTER_PK>(e, tt, sc, twr, FftBlockSync());
    }
    const int64_t bs = (int64_t)((r * BPR + (uint64_t)g) * (uint64_t)B) - p.goff;  // first output of the block
    if (interior(r)) {  // (uniform)
      c32* py = p.y + (bs + (t - ntm1));
#pragma unroll
      for (int m = 0; m < 8; m++)
        if (t + m * TF >= ntm1) py[m * TF] = e[m];
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const int64_t n = bs + (t + m * TF - ntm1);
        if (t + m * TF >= ntm1 && n >= 0 && (uint64_t)n < p.nsamples) p.y[n] = e[m];
      }
    }
  };
  // two windows take turns: while one is transformed the other one's loads are in flight (no register copy)
  c32 xa[8], xb[8];
  uint64_t r = blockIdx.x;
  const uint64_t stride = gridDim.x;
  if (r < p.nrounds) load_window(xa, r);
  for (; r < p.nrounds; r += 2 * stride) {
    if (r + stride < p.nrounds) load_window(xb, r + stride);
    do_round(xa, r);
    if (r + stride < p.nrounds) {
      if (r + 2 * stride < p.nrounds) load_window(xa, r + 2 * stride);
      do_round(xb, r + stride);
    }
  }
}

struct SyncLds {
  size_t ys, mt, mh, mh1, misc, total;
};
// LDS of k_sync: the ring of filtered samples | the tile's M values (mt) | the CP newest M values of the previous
// tile (mh), double-buffered: the next tile's are written while this tile's are still read | scan / vote scratch.
// Where the history in front of a tile is short (fixed layout, H = R - T <= 1000 samples: N <= 512), mt OVERLAYS the
// tile's own samples: every read of them precedes the barrier behind which M is written, the part the slide to the
// front still reads lies behind mt's end, and the next tile's samples arrive only after mt's last read -- 9 KB less per
// workgroup, which is what lets a CU hold five or six of them.
__host__ __device__ inline bool sync_mt_overlay(int R) { return R - SYNC_TILE <= 1000 && R - SYNC_TILE <= SYNC_TILE; }
__host__ __device__ inline SyncLds sync_lds_layout(int R, int HM) {
  SyncLds l;
  size_t o = 0;
  l.ys = o;
  o += (size_t)(sync_lp(R) + 2) * sizeof(c32);
  o = (o + 15) & ~(size_t)15;
  if (sync_mt_overlay(R)) {
    l.mt = l.ys + (size_t)sync_lp(R - SYNC_TILE) * sizeof(c32);  // the tile's first sample (a multiple of 8: 16-aligned)
  } else {
    l.mt = o;
    o += ((size_t)(sync_lp(SYNC_TILE) + 2) * sizeof(float) + 15) & ~(size_t)15;
  }
  l.mh = o;  // M history: the CP values before the tile
  o += ((size_t)(sync_lp(HM) + 2) * sizeof(float) + 15) & ~(size_t)15;
  l.mh1 = o;
  o += ((size_t)(sync_lp(HM) + 2) * sizeof(float) + 15) & ~(size_t)15;
  l.misc = o;
  o += 384;
  l.total = o;
  return l;
}
// LDS of the fused front end (k_sync<W, true, F>: channel filter + metric): [history | tile | carry] of filtered
// samples (carry: what the tile's last filter block produces beyond the tile's end, at most B - 1 samples) | the
// transforms' scratch, which the tile's M values (mt) take over between the filter phase and the next one | the M
// history, double-buffered (the next tile's is written while this tile's is still being read) | scan / vote scratch.
struct FrontLds {
  size_t ys, mt, mh0, mh1, tw, misc, total;
  int C;
};
__host__ __device__ inline FrontLds front_lds_layout(int R, int HM, int B, int F) {
  FrontLds l;
  l.C = (B + 7) / 8 * 8;
  size_t o = 0;
  l.ys = o;
  o += (size_t)(sync_lp(R + l.C) + 2) * sizeof(c32);
  o = (o + 15) & ~(size_t)15;
  l.mt = o;
  const size_t mtb = ((size_t)(sync_lp(SYNC_TILE) + 2) * sizeof(float) + 15) & ~(size_t)15;
  const size_t scb = (size_t)SYNC_THREADS * 9 * sizeof(c32);  // 256/(F/8) transforms x (F + F/8) points
  o += mtb > scb ? mtb : scb;
  l.mh0 = o;
  o += ((size_t)(sync_lp(HM) + 2) * sizeof(float) + 15) & ~(size_t)15;
  l.mh1 = o;
  o += ((size_t)(sync_lp(HM) + 2) * sizeof(float) + 15) & ~(size_t)15;
  l.tw = o;  // the filter transforms' twiddle table (fft.h FftTwLds): registers are what this kernel is short of
  o += ((size_t)fft_tw_lds_points(F) * sizeof(c32) + 15) & ~(size_t)15;
  l.misc = o;
  o += 384;
  l.total = o;
  return l;
}
// LDS of k_sync_exact: exact M over [amin-CP+1, bmax] (me) | exact u over [amin, bmax] (ue) | scan scratch
struct ExactLds {
  size_t me, ue, misc, total;
};
__host__ __device__ inline ExactLds exact_lds_layout(int CP, int rmax) {  // rmax: longest range handled
  ExactLds l;
  size_t o = 0;
  l.me = o;
  o += ((size_t)(rmax + CP + 8) * sizeof(float) + 15) & ~(size_t)15;
  l.ue = o;
  o += ((size_t)(rmax + 8) * sizeof(float) + 15) & ~(size_t)15;
  l.misc = o;
  o += 512;
  l.total = o;
  return l;
}

// slot of a ring-relative position s in (-R, 2R)
__device__ __forceinline__ int ring_wrap(int s, int R) {
  s += (s < 0) ? R : 0;
  s -= (s >= R) ? R : 0;
  return s;
}

// affine map a -> A*a + b, composition "first f then g"
struct Aff {
  double A, b;
};
__device__ __forceinline__ Aff aff_then(Aff f, Aff g) {
  Aff r;
  r.A = f.A * g.A;
  r.b = f.b * g.A + g.b;
  return r;
}

// ---- DPP wave scan (float): 4 row_shr steps inside rows of 16, then two row broadcasts ------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_incl_scan_f32(float v) {
  v += dpp_f<0x111, 0xF>(v);  // row_shr:1
  v += dpp_f<0x112, 0xF>(v);  // row_shr:2
  v += dpp_f<0x114, 0xF>(v);  // row_shr:4
  v += dpp_f<0x118, 0xF>(v);  // row_shr:8
  v += dpp_f<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

struct F3 {
  float a, b, c;
};

// The same network on several values at once, spelled as v_add_f32 with a DPP source (one instruction per step and
// value; from the builtin the compiler emits v_mov_b32_dpp + a packed add per pair of values plus the moves to pair
// them up: 89 instead of 36 instructions for the six scans of block_scan3_sum3).  A lane whose DPP source lies outside
// its row keeps its value (bound_ctrl off), exactly like `v += dpp_f(v)` above: same sums, bit for bit.  Inline asm is
// invisible to the hazard recogniser: a VGPR written by a VALU instruction needs two wait states before a DPP read, so
// consecutive steps of one value are kept three instructions apart (or separated by s_nop 1).
#define DPP_STEP3(CTRL, MASK)                                                        \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"          \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"          \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_incl_scan3_f32(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t" DPP_STEP3("row_shr:1", "0xf") DPP_STEP3("row_shr:2", "0xf") DPP_STEP3("row_shr:4", "0xf")
                   DPP_STEP3("row_shr:8", "0xf") DPP_STEP3("row_bcast:15", "0xa") DPP_STEP3("row_bcast:31", "0xc")
               : "+v"(a), "+v"(b), "+v"(c));
}
#define DPP_STEP2(CTRL, MASK)                                                        \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"          \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"          \
  "s_nop 0\n\t"
__device__ __forceinline__ void wave_incl_scan2_f32(float& a, float& b) {
  asm volatile("s_nop 1\n\t" DPP_STEP2("row_shr:1", "0xf") DPP_STEP2("row_shr:2", "0xf") DPP_STEP2("row_shr:4", "0xf")
                   DPP_STEP2("row_shr:8", "0xf") DPP_STEP2("row_bcast:15", "0xa") DPP_STEP2("row_bcast:31", "0xc")
               : "+v"(a), "+v"(b));
}
#define DPP_MAX1(CTRL, MASK) "v_max_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\ts_nop 1\n\t"
// the same network with max instead of add: inclusive running maximum over the lanes of a wave
__device__ __forceinline__ float wave_incl_scanmax_f32(float a) {
  asm volatile("s_nop 1\n\t" DPP_MAX1("row_shr:1", "0xf") DPP_MAX1("row_shr:2", "0xf") DPP_MAX1("row_shr:4", "0xf")
                   DPP_MAX1("row_shr:8", "0xf") DPP_MAX1("row_bcast:15", "0xa") DPP_MAX1("row_bcast:31", "0xc")
               : "+v"(a));
  return a;
}
#define DPP_STEP1(CTRL, MASK) "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\ts_nop 1\n\t"
__device__ __forceinline__ float wave_incl_scan1_f32(float a) {
  asm volatile("s_nop 1\n\t" DPP_STEP1("row_shr:1", "0xf") DPP_STEP1("row_shr:2", "0xf") DPP_STEP1("row_shr:4", "0xf")
                   DPP_STEP1("row_shr:8", "0xf") DPP_STEP1("row_bcast:15", "0xa") DPP_STEP1("row_bcast:31", "0xc")
               : "+v"(a));
  return a;
}

// Block-wide (4 waves) scans / sums.  Each call has ONE barrier; `scratch` must not be a region
// another call of the same tile iteration used since the last-but-one barrier (callers rotate
// three regions).
__device__ __forceinline__ void block_scan3_sum3(F3 v, F3 s, float* scratch, F3* excl, F3* sum) {
  const int lane = lane_id(), w = wave_id();
  F3 inc = v, rs = s;
  wave_incl_scan3_f32(inc.a, inc.b, inc.c);
  wave_incl_scan3_f32(rs.a, rs.b, rs.c);
  if (lane == WAVE - 1) {
    scratch[w * 6 + 0] = inc.a;
    scratch[w * 6 + 1] = inc.b;
    scratch[w * 6 + 2] = inc.c;
    scratch[w * 6 + 3] = rs.a;
    scratch[w * 6 + 4] = rs.b;
    scratch[w * 6 + 5] = rs.c;
  }
  __syncthreads();
  F3 base = {0.f, 0.f, 0.f}, sm = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SYNC_THREADS / WAVE; i++) {
    if (i < w) {
      base.a += scratch[i * 6 + 0];
      base.b += scratch[i * 6 + 1];
      base.c += scratch[i * 6 + 2];
    }
    sm.a += scratch[i * 6 + 3];
    sm.b += scratch[i * 6 + 4];
    sm.c += scratch[i * 6 + 5];
  }
  excl->a = base.a + inc.a - v.a;
  excl->b = base.b + inc.b - v.b;
  excl->c = base.c + inc.c - v.c;
  *sum = sm;
}
__device__ __forceinline__ void block_scan1_sum1(float v, float s, float* scratch, float* excl, float* sum) {
  const int lane = lane_id(), w = wave_id();
  float inc = v, rs = s;
  wave_incl_scan2_f32(inc, rs);
  if (lane == WAVE - 1) {
    scratch[w * 2 + 0] = inc;
    scratch[w * 2 + 1] = rs;
  }
  __syncthreads();
  float base = 0.f, sm = 0.f;
#pragma unroll
  for (int i = 0; i < SYNC_THREADS / WAVE; i++) {
    if (i < w) base += scratch[i * 2 + 0];
    sm += scratch[i * 2 + 1];
  }
  *excl = base + inc - v;
  *sum = sm;
}

struct Q3 {
  long long pr, pi, r;
};

// exclusive block scan of three int64 plus block sum of three more: one barrier pair.
// scratch: 4 waves x 6 int64 (rare path: plain shuffles)
template <int NT>
__device__ __forceinline__ void block_scan3_sum3_i64(Q3 v, Q3 s, long long* scratch, Q3* excl, Q3* sum) {
  const int lane = lane_id(), w = wave_id();
  Q3 inc = v, rs = s;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const long long a = __shfl_up(inc.pr, d, WAVE), b = __shfl_up(inc.pi, d, WAVE), c = __shfl_up(inc.r, d, WAVE);
    if (lane >= d) {
      inc.pr += a;
      inc.pi += b;
      inc.r += c;
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    rs.pr += __shfl_xor(rs.pr, d, WAVE);
    rs.pi += __shfl_xor(rs.pi, d, WAVE);
    rs.r += __shfl_xor(rs.r, d, WAVE);
  }
  if (lane == WAVE - 1) {
    scratch[w * 6 + 0] = inc.pr;
    scratch[w * 6 + 1] = inc.pi;
    scratch[w * 6 + 2] = inc.r;
    scratch[w * 6 + 3] = rs.pr;
    scratch[w * 6 + 4] = rs.pi;
    scratch[w * 6 + 5] = rs.r;
  }
  __syncthreads();
  Q3 base = {0, 0, 0}, sm = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < NT / WAVE; i++) {
    if (i < w) {
      base.pr += scratch[i * 6 + 0];
      base.pi += scratch[i * 6 + 1];
      base.r += scratch[i * 6 + 2];
    }
    sm.pr += scratch[i * 6 + 3];
    sm.pi += scratch[i * 6 + 4];
    sm.r += scratch[i * 6 + 5];
  }
  __syncthreads();
  excl->pr = base.pr + inc.pr - v.pr;
  excl->pi = base.pi + inc.pi - v.pi;
  excl->r = base.r + inc.r - v.r;
  *sum = sm;
}

// ---------------------------------------------------------------------------------
// Normative (Q23.40) evaluation of u = Mbar - 1 for the tile-relative samples [amin, bmax]:
//   me[k]  exact M of sample amin-CP+1+k        (k < bmax-amin+CP)
//   ue[k]  exact u of sample amin+k
//   gP[k]  exact P of sample amin+k, gU[k] its u: written straight to the candidate arrays (global)
// Rare path (only where the float32 pre-selection found something): kept out of line so that it
// does not weigh on the register allocation of the streaming loop.
// ---------------------------------------------------------------------------------
template <int NT, bool KEEP>
__device__ __forceinline__ void sync_exact_range(const c32* __restrict__ y, float* me, float* ue, c32* gP, float* gU, long long* sc_i64,
                                                 int amin, int bmax, int D, int CP, int64_t t0s, float tapcp) {
  const int tid = threadIdx.x;
  // One window term from its two samples (the sample and the one D before it): the correlator product and the energy
  // in Q23.40.  Samples before the stream start read as zero (a zero sample gives a zero term: the sample D before it
  // lies before the start as well).
  auto ldy = [&](int m) -> c32 {
    const int64_t ia = t0s + (int64_t)m;
    return ia >= 0 ? y[ia] : mk(0.f, 0.f);
  };
  auto qterm = [&](c32 a_, c32 d_) -> Q3 {
    const c32 c_ = cmul_conj(a_, d_);
    Q3 q_;
    q_.pr = q40_clamped(c_.re);
    q_.pi = q40_clamped(c_.im);
    q_.r = q40_clamped(a_.re * a_.re + a_.im * a_.im);
    return q_;
  };
  // The loops below fetch the samples of EXACT_G steps before using any of them: one round trip to L2 / HBM per group
  // instead of one per step (these loads and their waits were most of the kernel: a record is a chain of ~18 of them).
  constexpr int EXACT_G = 4;
  const int s0 = amin - CP;               // anchor sample (tile-relative, may be negative)
  const int n_e = bmax - s0;              // samples s0+1 .. bmax get an exact M
  const int len = bmax - amin + 1;
  // (i) exact window sums at the anchor
  Q3 an = {0, 0, 0};
  for (int m0 = s0 - D + 1 + tid; m0 <= s0; m0 += EXACT_G * NT) {
    c32 ya[EXACT_G], yd[EXACT_G];
#pragma unroll
    for (int u = 0; u < EXACT_G; u++) {
      const int m = m0 + u * NT;
      const bool on = m <= s0;
      ya[u] = on ? ldy(m) : mk(0.f, 0.f);
      yd[u] = on ? ldy(m - D) : mk(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < EXACT_G; u++) {
      if (m0 + u * NT <= s0) {
        const Q3 q = qterm(ya[u], yd[u]);
        an.pr += q.pr;
        an.pi += q.pi;
        an.r += q.r;
      }
    }
  }
  // (ii) per-thread chunk of the delta sequence, scan, exact M
  const int lc = (n_e + NT - 1) / NT;
  const int k0 = tid * lc, k1 = (k0 + lc < n_e) ? (k0 + lc) : n_e;
  // the deltas (new term minus the term leaving the window) of steps k .. k + EXACT_G - 1 of this thread's chunk
  auto deltas = [&](int k, Q3* d) {
    c32 ya[EXACT_G], yd[EXACT_G], ydd[EXACT_G];
#pragma unroll
    for (int u = 0; u < EXACT_G; u++) {
      const int m = s0 + 1 + k + u;
      const bool on = k + u < k1;
      ya[u] = on ? ldy(m) : mk(0.f, 0.f);
      yd[u] = on ? ldy(m - D) : mk(0.f, 0.f);
      ydd[u] = on ? ldy(m - 2 * D) : mk(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < EXACT_G; u++) {
      const Q3 a = qterm(ya[u], yd[u]), b = qterm(yd[u], ydd[u]);
      d[u].pr = a.pr - b.pr;
      d[u].pi = a.pi - b.pi;
      d[u].r = a.r - b.r;
    }
  };
  // A thread's deltas are needed twice -- for its chunk total before the scan and again, one by one, after it.  Chunks
  // of up to EXACT_KEEP samples keep them in registers between the two passes; longer chunks evaluate them again.
  // KEEP is a template parameter because the registers cost residency: measured, it pays at N = 4096 (k_sync_exact
  // 1.20 -> 0.89 ms at C5) and loses at N = 2048 / 512 (0.58 -> 0.63, 0.63 -> 0.69 ms): the host turns it on for
  // D >= 2048.
#ifndef EXACT_KEEP_N
#define EXACT_KEEP_N 12  // (a multiple of EXACT_G)
#endif
  constexpr int EXACT_KEEP = KEEP ? EXACT_KEEP_N : EXACT_G;
  const bool keep = KEEP && lc <= EXACT_KEEP;
  Q3 dq[EXACT_KEEP];
  Q3 tq = {0, 0, 0};
  if (keep) {
#pragma unroll
    for (int i = 0; i < EXACT_KEEP; i += EXACT_G) deltas(k0 + i, dq + i);
#pragma unroll
    for (int i = 0; i < EXACT_KEEP; i++) {
      if (k0 + i < k1) {
        tq.pr += dq[i].pr;
        tq.pi += dq[i].pi;
        tq.r += dq[i].r;
      }
    }
  } else {
    for (int k = k0; k < k1; k += EXACT_G) {
      Q3 d[EXACT_G];
      deltas(k, d);
#pragma unroll
      for (int u = 0; u < EXACT_G; u++) {
        if (k + u < k1) {
          tq.pr += d[u].pr;
          tq.pi += d[u].pi;
          tq.r += d[u].r;
        }
      }
    }
  }
  Q3 ex, ansum;
  block_scan3_sum3_i64<NT>(tq, an, sc_i64, &ex, &ansum);
  long long wpr = ansum.pr + ex.pr, wpi = ansum.pi + ex.pi, wr = ansum.r + ex.r;
  auto emit = [&](int k, Q3 d) {
    const int m = s0 + 1 + k;
    wpr += d.pr;
    wpi += d.pi;
    wr += d.r;
    const float pre = (float)q40_to_double(wpr);
    const float pim = (float)q40_to_double(wpi);
    const float r = (float)q40_to_double(wr);
    const float num = pre * pre + pim * pim;
    const float den = r * r;
    float mm = (den > 0.0f) ? (num / den) : 0.0f;
    if (!(mm <= 1024.0f)) mm = 1024.0f;
    me[k] = mm;
    if (gP && m >= amin) gP[m - amin] = mk(pre, pim);
  };
  if (keep) {
#pragma unroll
    for (int i = 0; i < EXACT_KEEP; i++)
      if (k0 + i < k1) emit(k0 + i, dq[i]);
  } else {
    for (int k = k0; k < k1; k += EXACT_G) {
      Q3 d[EXACT_G];
      deltas(k, d);
#pragma unroll
      for (int u = 0; u < EXACT_G; u++)
        if (k + u < k1) emit(k + u, d[u]);
    }
  }
  __syncthreads();
  // (iii) exact CP-length moving sum of M over [amin, bmax]; me[k] holds sample s0+1+k = amin-CP+1+k
  long long am = 0;
  for (int k = tid; k < CP; k += NT) am += q40_from_float(me[k]);  // samples amin-CP+1 .. amin
  const int lc2 = (len + NT - 1) / NT;
  const int j0 = tid * lc2, j1 = (j0 + lc2 < len) ? (j0 + lc2) : len;
  long long tm = 0;
  for (int jj = j0; jj < j1; jj++)
    if (jj > 0) tm += q40_from_float(me[CP - 1 + jj]) - q40_from_float(me[jj - 1]);
  Q3 mv = {tm, 0, 0}, ms_ = {am, 0, 0}, mex, msum;
  block_scan3_sum3_i64<NT>(mv, ms_, sc_i64, &mex, &msum);
  long long wm = msum.pr + mex.pr;
  for (int jj = j0; jj < j1; jj++) {
    if (jj > 0) wm += q40_from_float(me[CP - 1 + jj]) - q40_from_float(me[jj - 1]);
    const float mbar = (float)(q40_to_double(wm) * (double)tapcp);
    const float ux = mbar + (-1.0f);
    ue[jj] = ux;
    if (gU) gU[jj] = ux;
  }
  __syncthreads();
}

#ifdef SYNC_STAMPS
#define STAMP(i)                                                             \
  do {                                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                      \
    st_acc[i] += now_ - st_last;                                             \
    st_last = now_;                                                          \
  } while (0)
#define STAMP_VM(i)                                                          \
  do {                                                                       \
    __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */            \
    STAMP(i);                                                                \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_VM(i) do { } while (0)
#endif

// Phase ablation for timing experiments exists only in the diagnostic build (make diag ->
// libofdm_hip_diag.so, selected with OFDM_HIP_LIB); the product library always runs every phase.
#ifdef SYNC_DIAG
#define SYNC_ABLATE(p, bit) ((p).ablate & (bit))
#else
#define SYNC_ABLATE(p, bit) 0
#endif
// Thread 0 puts the waves' parts of a tile's detector summary together (see the end of k_sync's tile loop): either the
// float32 summary of a tile without candidates, or the record k_sync_exact works from.  A tile in which the float32
// window energy fell below SYNC_ILL of its running maximum (cancellation: the float32 metric cannot be trusted there) --
// or whose predecessor did, whose M values its first CP means still average -- is handed over whole.  Returns whether
// this tile was ill-conditioned.
__device__ __forceinline__ bool sync_tile_ill(const float* slots, uint64_t tile) {
  constexpr int NW = SYNC_THREADS / WAVE;
  const float* sl = slots + (int)(tile & 1u) * NW * 6;
  bool ill = false;
#pragma unroll
  for (int w = 0; w < NW; w++) ill = ill || reinterpret_cast<const int*>(sl + 6 * w)[5] != 0;
  return ill;
}
__device__ __forceinline__ bool sync_finish_tile(const SyncParams& p, const float* slots, uint64_t tile, bool prev_ill) {
  constexpr int NW = SYNC_THREADS / WAVE;
  const float* sl = slots + (int)(tile & 1u) * NW * 6;
  const bool ill = sync_tile_ill(slots, tile);
  if (ill || prev_ill) {
    const uint64_t t0 = tile * (uint64_t)SYNC_TILE;
    const unsigned long long r = atomicAdd(p.rec_count, 1ull);
    SyncRec rec;
    rec.tile = tile;
    rec.amin = 0;
    rec.bmax = ((t0 + (uint64_t)SYNC_TILE <= p.nsamples) ? SYNC_TILE : (int)(p.nsamples - t0)) - 1;
    rec.gpre = rec.gpost = 0.0f;
    p.recs[r] = rec;
    return ill;
  }
  int w0 = -1, w1 = -1;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    if (reinterpret_cast<const int*>(sl + 6 * w)[4] >= 0) {
      if (w0 < 0) w0 = w;
      w1 = w;
    }
  }
  if (w0 < 0) {
    float tb = sl[0];
    if (NW == 4) tb = (sl[0] + sl[6]) + (sl[12] + sl[18]);
    else if (NW == 2) tb = sl[0] + sl[6];
    p.tile_B[tile] = (double)tb;
    p.tile_npieces[tile] = 0;
    return ill;
  }
  float ga = 0.f, gb = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    ga += (w < w0) ? sl[6 * w] : (w == w0) ? sl[6 * w + 1] : 0.0f;
    gb += (w > w1) ? sl[6 * w] : (w == w1) ? sl[6 * w + 2] : 0.0f;
  }
  const unsigned long long r = atomicAdd(p.rec_count, 1ull);  // (capacity: one record per tile)
  SyncRec rec;
  rec.tile = tile;
  rec.amin = reinterpret_cast<const int*>(sl + 6 * w0)[3];
  rec.bmax = reinterpret_cast<const int*>(sl + 6 * w1)[4];
  rec.gpre = ga;
  rec.gpost = gb;
  p.recs[r] = rec;
  return ill;
}

// W: workgroups per CU the register allocation aims at -- 3 when the LDS footprint allows three, else 2 (long
// symbols: the y history alone is N+CP samples).
// STATIC (history no longer than a tile: N <= 1024 at the usual prefix lengths): the LDS holds [history | tile] at fixed
// places and the newest R - T samples are copied to the front after each tile's last read -- every LDS address of the
// tile loop is then one per-thread base plus uniform offsets, where the ring costs a wrap per access.
//
// NORMATIVE SCHEDULE.  The float32 values this kernel produces (u of every sample, the per-wave parts of a tile's
// detector summary) are part of the receiver's defined result: they pick the ranges of the fixed-point evaluation and
// feed the peak detector's average outside them, so the oracle performs the same operations in the same order
// (oracle/ofdm_oracle.c, peak_detect: "lanes" = threads, "scan network" = wave_incl_scan*_f32, "groups" = waves).
// A tile's values depend only on y[t0 - 2D - CP .. t0 + T): window sums anchored afresh at the tile start, the CP
// values of M before the tile computed by the previous tile's evaluation.  A segment therefore starts one tile early (the
// warm-up tile: a full evaluation whose M values seed the first owned tile, its u discarded) behind a prologue that
// loads the 2D samples of history -- any segmentation of the stream gives the same bits.
// F > 0: the FUSED front end -- the channel filter (k_chan_filter's overlap-save blocks, same transforms, same bits) runs
// inside this kernel: the filter blocks whose first output lies in a tile are transformed by the workgroup right before
// that tile's metric, their outputs going to HBM (y, for the demodulator and k_sync_exact) AND straight into the LDS
// window the metric reads -- the filtered stream is written once and not read back here (5.3 KB per symbol of HBM reads
// at C2).  A thread keeps its place in its filter block: twiddles and its bins of the transformed taps stay in registers
// across the metric phases; the next block's input window is in flight while the metric runs.  Only a segment's owner
// stores y (the prologue and the warm-up tile recompute what they need: blocks lie on the capture's grid, so the
// values are the same).  Needs the fixed LDS layout with history + carry no longer than a tile, and F <= 512 (a
// transform's threads inside one wave).
template <int W, bool STATIC, int F = 0>
__global__ void __launch_bounds__(SYNC_THREADS, W) k_sync(SyncParams p, FilterParams fp) {
#ifdef SYNC_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  constexpr bool FUSED = F > 0;
  static_assert(!FUSED || (STATIC && F <= 512), "the fused front end uses the fixed LDS layout and wave-sized transforms");
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  constexpr int T = SYNC_TILE;
  const int R = p.R;
  const SyncLds L0 = sync_lds_layout(R, p.HM);
  const FrontLds L1 = front_lds_layout(R, p.HM, FUSED ? fp.B : 8, FUSED ? F : 64);
  c32* ys = reinterpret_cast<c32*>(smem + (FUSED ? L1.ys : L0.ys));
  float* mh = reinterpret_cast<float*>(smem + (FUSED ? L1.mh0 : L0.mh));
  float* mh_nxt = reinterpret_cast<float*>(smem + (FUSED ? L1.mh1 : L0.mh1));  // the next tile's M history
  float* mt = reinterpret_cast<float*>(smem + (FUSED ? L1.mt : L0.mt));
  unsigned char* misc = smem + (FUSED ? L1.misc : L0.misc);
  float* scA = reinterpret_cast<float*>(misc);                 // 24 floats
  float* scB = reinterpret_cast<float*>(misc + 96);            // 8 floats

  const uint64_t seg = blockIdx.x;
  const uint64_t tile_own0 = seg * (uint64_t)p.tiles_per_seg;
  uint64_t tile_own1 = tile_own0 + (uint64_t)p.tiles_per_seg;
  if (tile_own1 > p.ntiles) tile_own1 = p.ntiles;
  const uint64_t tile_first = seg > 0 ? tile_own0 - 1 : tile_own0;
  const int D = p.D, CP = p.CP, HM = p.HM;
  const float inv_cp = 1.0f / (float)CP;
  const int H = R - T;                                     // history samples in front of a tile (>= 2D)

  // ---- the filter's side of the fused kernel ----------------------------------------------------------------
  // A tile needs at most two rounds of the workgroup's BPR blocks (B > F/2: fewer than 2 BPR blocks start in 2048
  // samples), so a thread keeps TWO input windows in registers across the metric phase -- fetched a whole tile ahead,
  // which is what hides HBM's loaded latency with only three workgroups per CU.  The twiddles come from a table in LDS
  // and the thread's bins of the transformed taps are re-read (L1 / L2) at the start of every filter phase, BEFORE the
  // prefetch in program order: nothing else may stay in registers over the metric phase at this budget.
  constexpr int FF = FUSED ? F : 64;      // (a legal length for the unfused instantiation's dead code)
  constexpr int TF = FF / 8;              // threads per filter block
  constexpr int BPR = SYNC_THREADS / TF;  // blocks per round of the workgroup
  const int fg = tid / TF, ft = tid % TF;
  c32* fsc = reinterpret_cast<c32*>(mt) + fg * fft_lds_points(FF);  // this block's transform scratch (overlays mt)
  c32* twl = reinterpret_cast<c32*>(smem + L1.tw);
  c32 xna[8], xnb[8];
  const int64_t own_lo = (int64_t)(tile_own0 * (uint64_t)T);
  const int64_t own_hi = (int64_t)((tile_own1 * (uint64_t)T < p.nsamples) ? tile_own1 * (uint64_t)T : p.nsamples);
  // first block with its first output at or behind sample n >= 0 (block b starts at b*B - goff)
  auto blk_at = [&](uint64_t n) -> int64_t { return (int64_t)((n + (uint64_t)fp.goff + (uint64_t)fp.B - 1) / (uint64_t)fp.B); };
  auto load_window = [&](c32 (&w)[8], int64_t b) {
    const int64_t x0 = b * fp.B - fp.goff - fp.ntm1 + ft;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int64_t xi = x0 + m * TF;
      w[m] = (xi >= 0 && (uint64_t)xi < p.nsamples) ? fp.x[xi] : mk(0.f, 0.f);
    }
  };
  // one overlap-save block: window -> transform -> x transformed taps -> inverse -> the last B points to HBM (owned
  // samples) and into the LDS window [history | tile | carry] of the tile that starts at t0s
  auto do_block = [&](const c32 (&w)[8], const c32 (&Hr)[8], int64_t b, int64_t t0s) {
    c32 e[8];
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = w[m];
    int tt = ft;  // opaque copy, renewed every block: keeps the transforms' LDS addresses out of the registers
    asm volatile("" : "+v"(tt));
    fft_run1<FF, false, FILTER_PK, FftWaveSync, FftTwLds>(e, tt, fsc, FftTwLds{twl}, FftWaveSync());
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = cmul(e[m], Hr[m]);  // volk_32fc_x2_multiply_32fc
    asm volatile("" : "+v"(tt));
    fft_run1<FF, true, FILTER_PK, FftWaveSync, FftTwLds>(e, tt, fsc, FftTwLds{twl}, FftWaveSync());
    const int64_t bs = b * fp.B - fp.goff;  // first output of the block
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int idx = ft + m * TF;
      const int64_t n = bs + (idx - fp.ntm1);
      const int64_t pos = n - t0s + H;  // place in [history | tile | carry]
      if (idx >= fp.ntm1 && n >= 0 && pos >= 0) {
        const bool in = (uint64_t)n < p.nsamples;
        if (!SYNC_ABLATE(p, 8)) ys[sync_lp((int)pos)] = in ? e[m] : mk(0.f, 0.f);  // (y behind the end of the stream reads as zero)
        if (n >= own_lo && n < own_hi && !SYNC_ABLATE(p, 4)) fp.y[n] = e[m];
      }
    }
  };
  auto load_taps = [&](c32 (&Hr)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) Hr[m] = fp.Hf[ft + m * TF];
  };
  int64_t bcur = 0;  // next filter block (fused)

  // ---- segment prologue: the y history of the first tile walked (y before the stream start reads as zero; so does the
  //      M history, which only the stream's first tile ever uses: a warm-up tile's u is discarded) ----
  if constexpr (FUSED) {
    for (int i = tid; i < fft_tw_used(FF); i += SYNC_THREADS) twl[lpad(i)] = fp.twF[i];
    for (int i = tid; i < sync_lp(R + L1.C) + 2; i += SYNC_THREADS) ys[i] = mk(0.f, 0.f);
    for (int i = tid; i < HM; i += SYNC_THREADS) mh_nxt[sync_lp(i)] = 0.0f;
    const uint64_t tf0 = tile_first * (uint64_t)T;
    const int64_t bp0 = tf0 + (uint64_t)fp.goff >= (uint64_t)H ? (int64_t)((tf0 + (uint64_t)fp.goff - (uint64_t)H) / (uint64_t)fp.B) : 0;
    const int64_t bp1 = blk_at(tf0);
    __syncthreads();  // (the zero fill and the twiddle table before the prologue's transforms)
    {
      // the blocks that cover the history (once per segment: plain loads, no prefetch)
      c32 Hr[8];
      load_taps(Hr);
      for (int64_t bb = bp0; bb < bp1; bb += BPR) {
        const int64_t b = bb + fg;
        if (b < bp1) {  // (a transform's threads share the decision: one block = TF consecutive threads)
          load_window(xna, b);
          do_block(xna, Hr, b, (int64_t)tf0);
        }
      }
    }
    bcur = bp1;
    // the first tile's windows
    const int64_t be = blk_at(tf0 + T);
    if (bcur + fg < be) load_window(xna, bcur + fg);
    if (bcur + BPR + fg < be) load_window(xnb, bcur + BPR + fg);
  } else {
    const int64_t h0 = (int64_t)(tile_first * (uint64_t)T) - H;
    for (int c = tid; c < H; c += SYNC_THREADS) {
      const int64_t n = h0 + c;
      const c32 v = n >= 0 ? p.y[n] : mk(0.f, 0.f);          // (n < nsamples: it lies before a tile that exists)
      ys[sync_lp(STATIC ? c : ring_wrap(c - H, R))] = v;
    }
    for (int i = tid; i < 2; i += SYNC_THREADS) ys[sync_lp(R) + i] = mk(0.f, 0.f);
  }
  for (int i = tid; i < HM; i += SYNC_THREADS) mh[sync_lp(i)] = 0.0f;
  const bool y_al16 = ((uintptr_t)p.y & 15) == 0;
  // weight of this thread's 8 samples in the tile summary of the detector average
  const float decay_f = (float)p.decay;
  const float wfull = p.wtab[tid];
  float* slots = reinterpret_cast<float*>(misc + 160);  // [2 tile parities][waves][6]: the waves' parts of a tile summary
  bool pend = false;       // a tile's summary waits to be put together (uniform)
  bool pend_warm = false;  // the warm-up tile's ill-conditioning flags wait to be read (uniform)
  bool prev_ill = false;   // (thread 0) the previous tile was ill-conditioned
  uint64_t pend_tile = 0;
  float* wmax = reinterpret_cast<float*>(misc + 352);  // [waves]: largest window energy of each wave's samples
  int rbase = STATIC ? R - SYNC_TILE : 0;  // the slot of the tile's first sample
  auto RW = [R](int s_) { return STATIC ? s_ : ring_wrap(s_, R); };
  // the next tile of y, fetched a tile ahead: its HBM latency hides behind the metric phase
  float4 ypre[SYNC_V / 2];
  bool have_pre = false;
  __syncthreads();

  for (uint64_t tile = tile_first; tile < tile_own1;
       tile++, rbase = STATIC ? rbase : ((rbase + T >= R) ? rbase + T - R : rbase + T)) {
    // Opaque copy of the thread index, renewed every tile: otherwise the compiler hoists every
    // tid-dependent LDS address of the loop body out of the loop and then spills them.
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const uint64_t t0 = tile * (uint64_t)T;
    const bool owned = tile >= tile_own0;
    int ncarry = 0;  // (fused) samples of the next tile this tile's last filter block has produced

    if constexpr (FUSED) {
      // ---- 1. the filter blocks whose first output lies in this tile: y to HBM and into the LDS window
      const int64_t b1 = blk_at(t0 + T);
      {
        c32 Hr[8];
        load_taps(Hr);  // (before the prefetch below in program order: its wait must not cover the next tile's windows)
        const int64_t ba = bcur + fg, bb = bcur + BPR + fg;
        STAMP_VM(10);  // (diagnostic build: the taps and the windows have arrived)
        if (ba < b1) do_block(xna, Hr, ba, (int64_t)t0);
        STAMP(8);
        if (bb < b1) do_block(xnb, Hr, bb, (int64_t)t0);
        STAMP(9);
      }
      bcur = b1;
      ncarry = (int)(b1 * fp.B - fp.goff - (int64_t)(t0 + T));
      if (tile + 1 < tile_own1) {  // the next tile's windows: in flight over this tile's metric phase
        const int64_t b2 = blk_at(t0 + 2ull * T);
        if (b1 + fg < b2) load_window(xna, b1 + fg);
        if (b1 + BPR + fg < b2) load_window(xnb, b1 + BPR + fg);
      }
    } else {
    // ---- 1. the tile of y into the ring.  Ring hazards: the slots written hold samples more than HY before this
    //         tile -- every read of them happened before the last barrier of the previous iteration.
    if (have_pre) {
#pragma unroll
      for (int r = 0; r < SYNC_V / 2; r++) {
        const int li = sync_lp(RW(rbase + 2 * (tl + r * SYNC_THREADS)));
        ys[li] = mk(ypre[r].x, ypre[r].y);
        ys[li + 1] = mk(ypre[r].z, ypre[r].w);
      }
    } else if (y_al16 && t0 + (uint64_t)T <= p.nsamples) {
      const float4* src = reinterpret_cast<const float4*>(p.y + t0);
#pragma unroll
      for (int r = 0; r < SYNC_V / 2; r++) {
        const float4 v = src[tl + r * SYNC_THREADS];
        const int li = sync_lp(RW(rbase + 2 * (tl + r * SYNC_THREADS)));
        ys[li] = mk(v.x, v.y);
        ys[li + 1] = mk(v.z, v.w);
      }
    } else {
#pragma unroll
      for (int r = 0; r < SYNC_V; r++) {
        const int i = tl + r * SYNC_THREADS;
        const uint64_t n = t0 + (uint64_t)i;
        c32 v = mk(0.f, 0.f);
        if (n < p.nsamples) v = p.y[n];
        ys[sync_lp(RW(rbase + i))] = v;
      }
    }
    have_pre = false;
    // (five and more workgroups per CU hide the tile's load latency among themselves: no register prefetch there --
    //  its sixteen registers are what that budget lacks)
    if (W <= 4 && tile + 1 < tile_own1 && y_al16 && t0 + 2ull * T <= p.nsamples) {
      const float4* src = reinterpret_cast<const float4*>(p.y + t0 + T);
#pragma unroll
      for (int r = 0; r < SYNC_V / 2; r++) ypre[r] = src[tl + r * SYNC_THREADS];
      have_pre = true;
    }
    }
    STAMP(1);
    __syncthreads();  // B2: the tile's y is in the ring
    STAMP(2);
    if (pend) {  // the previous tile's summary: every wave wrote its slot before this barrier
      if (tid == 0) prev_ill = pend_warm ? sync_tile_ill(slots, pend_tile) : sync_finish_tile(p, slots, pend_tile, prev_ill);
      pend = pend_warm = false;
    }
    const int ybs = RW(rbase + SYNC_V * tl);
    const int yb = sync_lp(ybs), yb1 = sync_lp(RW(ybs - D)), yb2 = sync_lp(RW(RW(ybs - D) - D));

    if (SYNC_ABLATE(p, 2)) {
      __syncthreads();
      continue;
    }
    // ---- 4. float32 Schmidl-Cox sums, anchored at the tile start --------------------------
    // (P as a packed pair per sample: the correlator product and the running sums are two v_pk_*_f32 each)
    cv pri[SYNC_V];
    float pfe[SYNC_V];
    F3 tsum = {0.f, 0.f, 0.f};
    F3 anc = {0.f, 0.f, 0.f};
    {
      cv tri = {0.f, 0.f};
      float te = 0.f;
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        const cv a = cv_of(ys[yb + j]);
        const cv d1 = cv_of(ys[yb1 + j]);
        const cv d2 = cv_of(ys[yb2 + j]);
        tri = tri + (pk_cmul_tw<true>(a, d1) - pk_cmul_tw<true>(d1, d2));  // a conj(d1) - d1 conj(d2)
        te += fmaf(a.x, a.x, a.y * a.y) - fmaf(d1.x, d1.x, d1.y * d1.y);
        pri[j] = tri;
        pfe[j] = te;
      }
      tsum.a = tri.x;
      tsum.b = tri.y;
      tsum.c = te;
      // anchor: the window sums at the sample before the tile, summed afresh from the history
      cv ari = {0.f, 0.f};
      for (int m = -D + tl; m < 0; m += SYNC_THREADS) {
        const int sa = RW(rbase + m);
        const cv a = cv_of(ys[sync_lp(sa)]);
        const cv d1 = cv_of(ys[sync_lp(RW(sa - D))]);
        ari = ari + pk_cmul_tw<true>(a, d1);
        anc.c += fmaf(a.x, a.x, a.y * a.y);
      }
      anc.a = ari.x;
      anc.b = ari.y;
    }
    STAMP(3);
    F3 ex3, anch;
    block_scan3_sum3(tsum, anc, scA, &ex3, &anch);  // B3
    STAMP(4);
    if constexpr (STATIC) {
      // every read of this tile's y happened before B3: slide the newest R - T samples to the front.  (Source and
      // destination do not overlap, R - T <= T; the next tile's store comes after the barriers below.)
      for (int c = tl; c < R - T + ncarry; c += SYNC_THREADS) ys[sync_lp(c)] = ys[sync_lp(T + c)];
    }
    float Mv[SYNC_V];
    const int mb = sync_lp(SYNC_V * tl);
    cv bri;  // window sums at the sample before this thread's first
    bri.x = anch.a + ex3.a;
    bri.y = anch.b + ex3.b;
    const float be = anch.c + ex3.c;
    float rlo = INFINITY, rhi = -INFINITY;  // smallest / largest window energy among this thread's samples
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) {
      const cv pq = bri + pri[j];
      const float r = be + pfe[j];
      rlo = fminf(rlo, r);
      rhi = fmaxf(rhi, r);
      const float num = fmaf(pq.x, pq.x, pq.y * pq.y);
      // IEEE division (correctly rounded: part of the normative schedule -- a hardware reciprocal has no bits another
      // machine can reproduce).  R = 0 means P = 0 too: the floor on the denominator gives the 0/0 -> 0 of the normative
      // metric without a compare and select.
      float m = num / fmaxf(r * r, 1e-37f);
      m = fminf(m, 1024.0f);  // (NaN -- Inf/Inf on garbage input -- goes to 1024 as well: fminf returns the number)
      Mv[j] = m;
    }
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) mt[mb + j] = Mv[j];
    // running maximum of the window energy over the tile (for the ill-conditioning test below)
    const float rinc = wave_incl_scanmax_f32(rhi);
    if (lane_id() == WAVE - 1) wmax[wave_id()] = rinc;
    __syncthreads();  // B4
    STAMP(5);

    // ---- 5. CP-length moving average of M (float32), minus one -----------------------------
    float pm[SYNC_V];
    float msum = 0.f;
    // M[n - CP] of sample i of the tile sits at position i of the sequence [history (HM = CP) | tile]
    if ((CP & 7) == 0) {  // a thread's 8 values come from one array, at constant offsets from one base
      const float* srcm = (SYNC_V * tl < HM) ? (mh + sync_lp(SYNC_V * tl)) : (mt + sync_lp(SYNC_V * tl - HM));
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        msum += Mv[j] - srcm[j];
        pm[j] = msum;
      }
    } else {
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        const int i = SYNC_V * tl + j;
        msum += Mv[j] - ((i < HM) ? mh[sync_lp(i)] : mt[sync_lp(i - HM)]);
        pm[j] = msum;
      }
    }
    float manc = 0.f;
    for (int m = -CP + tl; m < 0; m += SYNC_THREADS) manc += mh[sync_lp(HM + m)];
    // the CP newest M values become the next tile's history -- into the OTHER history buffer, before B5: behind it mt
    // belongs to the next tile's samples (or, fused, to the transforms' scratch) again
    for (int i = tl; i < HM; i += SYNC_THREADS) mh_nxt[sync_lp(i)] = mt[sync_lp(T - HM + i)];
    float mex, mach;
    block_scan1_sum1(msum, manc, scB, &mex, &mach);  // B5
    STAMP(6);
    {
      float* sw_ = mh;
      mh = mh_nxt;
      mh_nxt = sw_;
    }
    // Float32 has lost the window energy where it fell below SYNC_ILL of the largest value the running sums went
    // through since the tile's anchor: such a tile goes to the fixed-point evaluation whole (sync_finish_tile).
    float pmx = fmaxf(anch.c, rinc);
#pragma unroll
    for (int i = 0; i < SYNC_THREADS / WAVE - 1; i++)
      if (i < wave_id()) pmx = fmaxf(pmx, wmax[i]);
    const bool ill_lane = rlo < SYNC_ILL * pmx;
    if (!owned) {  // warm-up tile: only the histories matter -- and whether it was ill-conditioned
      const unsigned long long ib = __ballot(ill_lane);
      if (lane_id() == WAVE - 1)
        reinterpret_cast<int*>(slots + ((int)(tile & 1u) * (SYNC_THREADS / WAVE) + wave_id()) * 6)[5] = ib != 0ull;
      pend = pend_warm = true;
      pend_tile = tile;
      continue;
    }

    float u[SYNC_V];
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) u[j] = (mach + mex + pm[j]) * inv_cp - 1.0f;
    if (p.presel_tap) {
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        const uint64_t n = t0 + (uint64_t)(SYNC_V * tl + j);
        if (n < p.nsamples) p.presel_tap[n] = u[j];
      }
    }

    // ---- 6. tile summary of the detector's running average; float32 pre-selection --------------
    int nv = 0;
    unsigned amask = 0;  // approximate candidates
    float floc = 0.f;
    const float athr = p.cand_thr - SYNC_GUARD;
    if (t0 + (uint64_t)T <= p.nsamples) {
      nv = SYNC_V;
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        floc = fmaf(floc, decay_f, p.alpha * u[j]);
        amask |= (u[j] > athr) ? (1u << j) : 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        const uint64_t n = t0 + (uint64_t)(SYNC_V * tl + j);
        if (n < p.nsamples) {
          nv++;
          floc = fmaf(floc, decay_f, p.alpha * u[j]);
          if (u[j] > athr) amask |= 1u << j;
        }
      }
    }
    float wgt = wfull;
    if (t0 + (uint64_t)T > p.nsamples) {
      // the (short) last tile: weight by the samples that follow this thread's
      const int64_t after = (int64_t)(p.nsamples - t0) - (int64_t)(SYNC_V * tl + nv);
      wgt = (float)p.dpow[after > 0 ? after : 0];
    }
    // Each WAVE leaves its part of the tile's summary in an LDS slot and moves on -- no workgroup barrier here; thread 0
    // puts the parts together after the next barrier the loop meets anyway (B2 of the next tile, or the one after the
    // loop).  A wave's part: the weighted sum S of its samples; whether it saw a candidate; and if so the first / last
    // candidate position and the sums of its samples before the first (Pre) / after the last (Post).  The tile's range
    // is [first of the first such wave, last of the last], the float32 summary before it S of the earlier waves + Pre,
    // after it Post + S of the later waves -- added in wave order, the same additions as a block-wide scan would do.
    {
      float sw = wave_incl_scan1_f32(floc * wgt);
      const unsigned long long ib = __ballot(ill_lane && nv > 0);
      const unsigned long long cb = __ballot(amask != 0);
      int amin_w = T, bmax_w = -1;
      float pre_w = 0.f, post_w = 0.f;
      if (cb != 0ull) {  // (wave-uniform)
        const int l0 = __ffsll((long long)cb) - 1, l1 = 63 - __clzll((long long)cb);
        const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)amask, l0);
        const unsigned m1 = (unsigned)__builtin_amdgcn_readlane((int)amask, l1);
        const int wbase_l = tl - lane_id();  // first thread of this wave
        amin_w = SYNC_V * (wbase_l + l0) + __ffs((int)m0) - 1;
        bmax_w = SYNC_V * (wbase_l + l1) + 31 - __clz((int)m1);
        float fpre = 0.f, fpost = 0.f;
#pragma unroll
        for (int j = 0; j < SYNC_V; j++) {
          if (j < nv) {
            const int i = SYNC_V * tl + j;
            fpre = fmaf(fpre, decay_f, p.alpha * ((i < amin_w) ? u[j] : 0.0f));
            fpost = fmaf(fpost, decay_f, p.alpha * ((i > bmax_w) ? u[j] : 0.0f));
          }
        }
        pre_w = fpre * wgt;
        post_w = fpost * wgt;
        wave_incl_scan2_f32(pre_w, post_w);
      }
      if (lane_id() == WAVE - 1) {
        float* sl = slots + ((int)(tile & 1u) * (SYNC_THREADS / WAVE) + wave_id()) * 6;
        sl[0] = sw;
        sl[1] = pre_w;
        sl[2] = post_w;
        reinterpret_cast<int*>(sl)[3] = amin_w;
        reinterpret_cast<int*>(sl)[4] = bmax_w;
        reinterpret_cast<int*>(sl)[5] = ib != 0ull;
      }
    }
    pend = true;
    pend_tile = tile;
    STAMP(7);
  }
  if (pend && !pend_warm) {
    __syncthreads();
    if (tid == 0) sync_finish_tile(p, slots, pend_tile, prev_ill);
  }
#ifdef SYNC_STAMPS
  if (p.stamps && threadIdx.x == 0)
    for (int i = 0; i < 16; i++) p.stamps[blockIdx.x * 16 + i] = st_acc[i];
#endif
}

// ---------------------------------------------------------------------------------
// k_sync_exact: the tiles k_sync found something near the threshold in (about one in five at C2).  One workgroup per
// record, records taken round-robin from the list (its length is read on the device: no host round trip):
//   7. fixed-point (Q23.40, normative) re-evaluation of u and P over [amin, bmax], y read straight from HBM / L2;
//      the whole range becomes the tile's candidate values;
//   8. candidate pieces = maximal runs of exact u > theta, each with the tile-local detector average before it; the
//      tile's detector summary from the values just used: exact u inside the range, k_sync's float32 partial sums
//      outside.  (Where the window energy drops by 50-60 dB inside a tile the float32 sums lose the small R to
//      cancellation -- M is then large AND a few percent off, enough to move the average later tiles inherit.)
// ---------------------------------------------------------------------------------
// Two sizes of workgroup share the list: ranges of at most p.exact_small samples -- a preamble's plateau (about CP
// samples plus the guard band's margin): nearly all of them -- are one WAVE's work (64 threads: every scan
// wave-local, no barrier at all, 9 KB of LDS at C2, 17 workgroups per CU); longer ones (a carrier, the metric tap's
// whole tiles) take a 256-thread workgroup.
// (measured: at N = 2048 / 4096 a single wave per range is 3-4x slower than the workgroup -- the window sums alone are
//  N/2 terms per range -- so the wave-sized variant serves N <= 512 only: C2 1.15 -> 0.62 ms, C3 0.58 vs 1.5-2.4 ms)
__host__ __device__ inline int sync_exact_small(int CP, int D) {
  if (D > 256) return 0;
  const int r = 2 * CP + 768;
  return r > SYNC_TILE ? SYNC_TILE : r;
}
template <int NT, bool KEEP>
__global__ void __launch_bounds__(NT) k_sync_exact(SyncParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int T = SYNC_TILE;
  constexpr bool SMALL = NT == WAVE;
  constexpr unsigned CHUNK_C = SMALL ? 2048u : (unsigned)SYNC_CHUNK_C;  // candidates per allocation chunk (>= the longest range)
  const int tl = threadIdx.x;
  const ExactLds L = exact_lds_layout(p.CP, SMALL ? p.exact_small : T);
  float* me = reinterpret_cast<float*>(smem + L.me);
  float* ue = reinterpret_cast<float*>(smem + L.ue);
  unsigned char* misc = smem + L.misc;
  long long* sc_i64 = reinterpret_cast<long long*>(misc);      // 24 entries
  unsigned long long* bc = reinterpret_cast<unsigned long long*>(misc + 320);  // 2 broadcast words (chunk allocation)
  const int D = p.D, CP = p.CP;
  // the metric tap's pass walks every tile over its whole length and writes nothing but the tap
  const unsigned long long nrec = p.tap_only ? (unsigned long long)p.ntiles : *p.rec_count;
  // current allocation chunks of this workgroup (uniform across the block)
  unsigned long long cand_base = 0, piece_base = 0;
  uint32_t cand_left = 0, piece_left = 0;
  const double alpha_d = (double)p.alpha;

  for (unsigned long long ri = blockIdx.x; ri < nrec; ri += gridDim.x) {
    SyncRec rec;
    if (p.tap_only) {
      rec.tile = ri;
      rec.amin = 0;
      const uint64_t left = p.nsamples - ri * (uint64_t)T;
      rec.bmax = (left < (uint64_t)T ? (int)left : T) - 1;
      rec.gpre = rec.gpost = 0.f;
    } else {
      rec = p.recs[ri];
    }
    const uint64_t tile = rec.tile;
    const int amin = rec.amin, bmax = rec.bmax;
    if (!p.tap_only && (bmax - amin + 1 <= p.exact_small) != SMALL) continue;  // the other launch's record
    const uint64_t t0 = tile * (uint64_t)T;
    const int64_t t0s = (int64_t)t0;
    const int Tl = (t0 + (uint64_t)T <= p.nsamples) ? T : (int)(p.nsamples - t0);  // samples of this tile
    const int rlen = bmax - amin + 1;

    if (p.tap_only) {
      sync_exact_range<NT, KEEP>(p.y, me, ue, nullptr, p.metric_tap + t0 + (uint64_t)amin, sc_i64, amin, bmax, D, CP, t0s, p.tapcp);
      continue;  // (sync_exact_range ends with a barrier: me / ue / scratch are free again)
    }

    // ---- 7. fixed-point re-evaluation of [amin, bmax]; stored as this tile's candidate values (u, P) ----
    if ((uint32_t)rlen > cand_left) {
      if (tl == 0) bc[0] = atomicAdd(p.cand_count, (unsigned long long)CHUNK_C);  // fresh chunk
      __syncthreads();
      cand_base = bc[0];
      cand_left = CHUNK_C;
      __syncthreads();
    }
    const unsigned long long cbase = cand_base;
    bool fits = cbase + (unsigned long long)rlen <= p.cand_cap;
    cand_base += (unsigned long long)rlen;
    cand_left -= (uint32_t)rlen;
    if (fits) {
      sync_exact_range<NT, KEEP>(p.y, me, ue, p.cand_P + cbase, p.cand_u + cbase, sc_i64, amin, bmax, D, CP, t0s, p.tapcp);
    } else {
      for (int i = tl; i < rlen; i += NT) ue[i] = -1.0f;  // nothing can be stored: no candidates
      __syncthreads();
    }

    // ---- 8. pieces and summary: each thread walks lc consecutive samples of the range.  The range's part of the
    //      detector average is a sum of Q40-rounded float64 products alpha * u[k] * decay^(Tl-1-k) (weights to the
    //      tile's end): integer addition has no order, so this scan, the oracle's running sum and any other split
    //      of the range give the same bits.
    const int lc = (rlen + NT - 1) / NT;
    const int j0 = tl * lc, j1 = (j0 + lc < rlen) ? (j0 + lc) : rlen;
    long long xq = 0;
    int nstart = 0;
    // (the weights of four steps are fetched together: one round trip to the table per group, not per step)
    for (int kg = j0; kg < j1; kg += 4) {
      double dw[4];
#pragma unroll
      for (int u = 0; u < 4; u++) dw[u] = (kg + u < j1) ? p.dpow[Tl - 1 - (amin + kg + u)] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = kg + u;
        if (k < j1) {
          const float ux = ue[k];
          xq += q40_from_double((alpha_d * (double)ux) * dw[u]);
          if (ux > p.cand_thr && !(k > 0 && ue[k - 1] > p.cand_thr)) nstart++;
        }
      }
    }
    Q3 v3 = {xq, (long long)nstart, 0}, ex, tot;
    block_scan3_sum3_i64<NT>(v3, v3, sc_i64, &ex, &tot);
    const int npieces = (int)tot.pi;
    if ((uint32_t)npieces > piece_left) {
      if (tl == 0) bc[1] = atomicAdd(p.piece_count, (unsigned long long)SYNC_CHUNK_P);  // fresh chunk
      __syncthreads();
      piece_base = bc[1];
      piece_left = SYNC_CHUNK_P;
    }
    const unsigned long long basep = piece_base;
    fits = fits && (basep + (unsigned long long)npieces <= p.piece_cap);
    if (!fits && tl == 0) atomicOr(p.overflow, 1u);
    if (fits && npieces > 0) {
      long long X = ex.pr;  // the range's samples before this thread's first
      int so = (int)ex.pi;
      for (int kg = j0; kg < j1; kg += 4) {
        double dw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) dw[u] = (kg + u < j1) ? p.dpow[Tl - 1 - (amin + kg + u)] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = kg + u;
          if (k < j1) {
            const float ux = ue[k];
            const bool cand = ux > p.cand_thr;
            const uint64_t n = t0 + (uint64_t)(amin + k);
            if (cand && !(k > 0 && ue[k - 1] > p.cand_thr)) {
              SyncPiece* pc = p.pieces + basep + so;
              pc->start = n;
              pc->val_off = cbase + (unsigned long long)k;
              // the average (zero at the tile start) just before this sample: everything before it, weighted to the
              // tile's end, carried back by 1 / decay^(Tl - s)
              pc->bloc = ((double)rec.gpre + (double)X * Q40_INV) * p.ipow[Tl - (amin + k)];
              so++;
            }
            if (cand && !(k + 1 < rlen && ue[k + 1] > p.cand_thr)) p.pieces[basep + so - 1].end = n;
            X += q40_from_double((alpha_d * (double)ux) * dw[u]);
          }
        }
      }
    }
    if (tl == 0) {
      p.tile_B[tile] = ((double)rec.gpre + (double)tot.pr * Q40_INV) + (double)rec.gpost;
      p.tile_npieces[tile] = fits ? (uint32_t)npieces : 0u;
      p.tile_first[tile] = basep;
    }
    if (fits) {
      piece_base += (unsigned long long)npieces;
      piece_left -= (uint32_t)npieces;
    }
    __syncthreads();  // ue / me / scratch are reused by the next record
  }
}

// ---------------------------------------------------------------------------------
// avg carry-in per tile: avg_in[g] = running average just before the tile's first sample
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_avg_carry(const double* __restrict__ tile_B, uint64_t ntiles, uint64_t nsamples,
                                                    const double* __restrict__ dpow, double* __restrict__ avg_in) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ntiles) return;
  // every earlier tile is full (only the last tile of the stream can be short); normative: k ascending, product and sum
  // separately rounded, cut-off where the weight has dropped below 1e-30
  const double A = dpow[SYNC_TILE];
  double acc = 0.0, w = 1.0;
  for (uint64_t k = 1; k <= g; k++) {
    acc += w * tile_B[g - k];
    w *= A;
    if (w < 1e-30) break;
  }
  // the initial condition avg[-1] = 0 contributes nothing
  avg_in[g] = acc;
}

// ---------------------------------------------------------------------------------
// gr_peak_detector_fb on the candidate intervals.  One thread per tile walks the intervals
// that START in its tile (an interval may run on through following tiles).
// ---------------------------------------------------------------------------------
struct PeakParams {
  uint64_t ntiles, nsamples;
  float rise, fall, alpha;
  const double* dpow;  // decay^j, j = 0..SYNC_TILE (SyncParams)
  const uint32_t* tile_npieces;
  const uint64_t* tile_first;
  const SyncPiece* pieces;
  const double* avg_in;
  const float* cand_u;
  const c32* cand_P;
  uint32_t* counts;         // [ntiles] flags raised by intervals starting in the tile
  const uint32_t* offsets;  // exclusive scan of counts (write pass)
  uint64_t* peaks;          // [npeaks]
  c32* peak_P;              // [npeaks]
  // count pass: the first PEAK_STASH flags of every tile are kept so that the write pass is a plain copy
  // (k_peak_compact); a tile with more sets *stash_overflow and the state machine runs a second time
  uint64_t* stash_peaks;    // [ntiles][PEAK_STASH]
  c32* stash_P;             // [ntiles][PEAK_STASH]
  unsigned int* stash_overflow;
};
#define PEAK_STASH 4

template <bool WRITE>
__global__ void __launch_bounds__(256) k_peak(PeakParams p) {
  const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g0 >= p.ntiles) return;
  const uint32_t np0 = p.tile_npieces[g0];
  uint32_t nflag = 0;
  const uint32_t wbase = WRITE ? p.offsets[g0] : 0;
  const float one_m_alpha = 1.0f - p.alpha;
  for (uint32_t slot = 0; slot < np0; slot++) {
    uint64_t g = g0;
    SyncPiece pc = p.pieces[p.tile_first[g] + slot];
    // a piece that starts on the tile's first sample continues the previous tile's last piece
    if (slot == 0 && g > 0 && pc.start == g * (uint64_t)SYNC_TILE) {
      const uint32_t npp = p.tile_npieces[g - 1];
      if (npp > 0 && p.pieces[p.tile_first[g - 1] + npp - 1].end + 1 == pc.start) continue;
    }
    float avg = (float)(p.avg_in[g] * p.dpow[pc.start - g * (uint64_t)SYNC_TILE] + pc.bloc);
    int state = 0;
    float peak_val = -INFINITY;
    uint64_t peak_ind = 0;
    uint64_t peak_off = 0;  // where the peak's P sits in the candidate arrays (read only when a flag is raised)
    // the pending peak lies in the piece being walked, pk_k values in (its sample index and candidate offset are
    // formed when the piece ends or a flag is raised: no 64-bit bookkeeping per value)
    bool pk_cur = false;
    uint32_t pk_k = 0;
    bool open_at_stream_end = false;
    for (;;) {
      const uint32_t len = (uint32_t)(pc.end - pc.start + 1);  // a piece lies inside one tile
      const float* __restrict__ up = p.cand_u + pc.val_off;
      // gr_peak_detector_fb's automaton on one value, without branches on the common paths:
      //   searching: u > avg*rise opens a run, and the value is then looked at as the run's first;
      //   in a run: a new maximum is recorded; else u > avg*fall keeps the run; else the run ends -- a flag at
      //   the recorded maximum -- and the SAME value is looked at again by the searching detector.
      // The average moves exactly once per value on every path.
      auto step = [&](const float u, const uint32_t k) __attribute__((always_inline)) {
        bool s1 = (state != 0) || (u > avg * p.rise);
        bool newpk = s1 && (u > peak_val);
        if (s1 && !newpk && !(u > avg * p.fall)) {  // the run ends here (once per run)
          const uint64_t ind = pk_cur ? pc.start + pk_k : peak_ind;
          const uint64_t off = pk_cur ? pc.val_off + pk_k : peak_off;
          if (WRITE) {
            p.peaks[wbase + nflag] = ind;
            p.peak_P[wbase + nflag] = p.cand_P[off];
          } else if (nflag < PEAK_STASH) {
            p.stash_peaks[g0 * PEAK_STASH + nflag] = ind;
            p.stash_P[g0 * PEAK_STASH + nflag] = p.cand_P[off];
          }
          nflag++;
          peak_val = -INFINITY;
          pk_cur = false;
          s1 = u > avg * p.rise;
          newpk = s1 && (u > peak_val);
        }
        peak_val = newpk ? u : peak_val;
        pk_k = newpk ? k : pk_k;
        pk_cur = pk_cur || newpk;
        avg = p.alpha * u + one_m_alpha * avg;
        state = s1 ? 1 : 0;
      };
      // eight values per trip (the loads are independent of the state machine: issued together); whole trips carry
      // no per-value bounds test, the piece's last values are taken one by one
      uint32_t kb = 0;
      for (; kb + 8 <= len; kb += 8) {
        float ub[8];
#pragma unroll
        for (int e = 0; e < 8; e++) ub[e] = up[kb + e];
#pragma unroll
        for (int e = 0; e < 8; e++) step(ub[e], kb + e);
      }
      for (; kb < len; kb++) step(up[kb], kb);
      if (pk_cur) {
        peak_ind = pc.start + pk_k;
        peak_off = pc.val_off + pk_k;
        pk_cur = false;
      }
      // does the interval continue in the next tile?
      const uint64_t gn = g + 1;
      if (pc.end + 1 == gn * (uint64_t)SYNC_TILE && gn < p.ntiles && p.tile_npieces[gn] > 0) {
        const SyncPiece nx = p.pieces[p.tile_first[gn]];
        if (nx.start == pc.end + 1) {
          pc = nx;
          g = gn;
          continue;
        }
      }
      if (pc.end + 1 >= p.nsamples) open_at_stream_end = true;
      break;
    }
    // the sample after the interval has u <= theta: it closes an open run (unless the stream ended)
    if (state == 1 && !open_at_stream_end) {
      if (WRITE) {
        p.peaks[wbase + nflag] = peak_ind;
        p.peak_P[wbase + nflag] = p.cand_P[peak_off];
      } else if (nflag < PEAK_STASH) {
        p.stash_peaks[g0 * PEAK_STASH + nflag] = peak_ind;
        p.stash_P[g0 * PEAK_STASH + nflag] = p.cand_P[peak_off];
      }
      nflag++;
    }
  }
  if (!WRITE) {
    p.counts[g0] = nflag;
    if (nflag > PEAK_STASH) atomicOr(p.stash_overflow, 1u);
  }
}

// write pass when no tile raised more than PEAK_STASH flags: copy the stashed flags to their scanned places
__global__ void __launch_bounds__(256) k_peak_compact(PeakParams p) {
  const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g0 >= p.ntiles) return;
  const uint32_t n = p.counts[g0];
  if (n == 0) return;
  const uint32_t wbase = p.offsets[g0];
  for (uint32_t i = 0; i < n && i < PEAK_STASH; i++) {
    p.peaks[wbase + i] = p.stash_peaks[g0 * PEAK_STASH + i];
    p.peak_P[wbase + i] = p.stash_P[g0 * PEAK_STASH + i];
  }
}

// ---------------------------------------------------------------------------------
// device-wide exclusive scan (three small kernels); n up to 1024*1024*1024 elements
// ---------------------------------------------------------------------------------
#define SCAN_BLOCK 1024  // elements per block (256 threads x 4)

template <typename T>
__global__ void __launch_bounds__(256) k_scan_partials(const T* __restrict__ in, uint64_t n, T* __restrict__ partial) {
  __shared__ T sc[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
  T s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n) s += in[base + k];
  T tot;
  (void)block_excl_scan_add<T>(s, sc, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single block: exclusive scan of `partial` in place; total -> *total_out
template <typename T>
__global__ void __launch_bounds__(256) k_scan_top(T* __restrict__ partial, uint64_t nb, T* __restrict__ total_out) {
  __shared__ T sc[8];
  T carry = 0;
  for (uint64_t base = 0; base < nb; base += 256) {
    const uint64_t i = base + threadIdx.x;
    const T v = (i < nb) ? partial[i] : (T)0;
    T tot;
    const T ex = block_excl_scan_add<T>(v, sc, &tot);
    if (i < nb) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T>
__global__ void __launch_bounds__(256) k_scan_final(const T* __restrict__ in, uint64_t n, const T* __restrict__ partial,
                                                     T* __restrict__ out) {
  __shared__ T sc[8];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
  T v[4];
  T s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    v[k] = (base + k < n) ? in[base + k] : (T)0;
    s += v[k];
  }
  T tot;
  T ex = block_excl_scan_add<T>(s, sc, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
}
