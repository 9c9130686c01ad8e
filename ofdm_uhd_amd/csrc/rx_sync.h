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
#pragma once
#include "common.h"

#define SYNC_THREADS 256
#define SYNC_V 8                            // consecutive samples per thread
#define SYNC_TILE (SYNC_THREADS * SYNC_V)   // 2048 samples per tile

struct SyncPiece {
  uint64_t start;    // absolute sample index of the first candidate of the piece
  uint64_t end;      // absolute index of its last candidate
  uint64_t val_off;  // offset of its first sample in the candidate value arrays
  double bloc;       // zero-initialised running average over the tile's samples before `start`
};

struct SyncParams {
  int N, D, CP;
  int HX;         // x history kept in LDS (padded tap count, multiple of 8)
  int HY;         // y history (2*D)
  int HM;         // M history (CP)
  int ntaps_pad;  // multiple of 8
  int tiles_per_seg, nwarm;
  uint64_t nsamples, ntiles;
  float tapcp;       // float(1/CP)
  float cand_thr;    // -max(rise, fall)
  float alpha;       // peak detector alpha
  double decay;      // double(1.0f - alpha)
  const c32* x;
  c32* y;
  const float* taps;
  float* metric_tap;  // optional [nsamples]
  // outputs
  double* tile_B;          // [ntiles] zero-init running average over the tile
  uint32_t* tile_npieces;  // [ntiles] candidate pieces of the tile ...
  uint64_t* tile_first;    // [ntiles] ... stored at pieces[tile_first .. +tile_npieces)
  SyncPiece* pieces;       // [piece_cap]
  uint64_t piece_cap;
  unsigned long long* piece_count;  // device counter
  float* cand_u;           // [cand_cap]
  c32* cand_P;             // [cand_cap]
  uint64_t cand_cap;
  unsigned long long* cand_count;  // device counter
  unsigned int* overflow;          // device flag
};

__host__ __device__ inline int sync_lp(int i) { return i + (i >> 3); }

__host__ inline size_t sync_lds_bytes(const SyncParams& p) {
  size_t xs = (size_t)(sync_lp(p.HX + SYNC_TILE) + 2) * sizeof(c32);
  size_t ys = (size_t)(sync_lp(p.HY + SYNC_TILE) + 2) * sizeof(c32);
  size_t ms = (size_t)(sync_lp(p.HM + SYNC_TILE) + 2) * sizeof(float);
  size_t misc = 1024;
  return xs + ys + ((ms + 15) & ~(size_t)15) + misc;
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

__global__ void __launch_bounds__(SYNC_THREADS) k_sync(SyncParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x;
  const int T = SYNC_TILE;
  c32* xs = reinterpret_cast<c32*>(smem);
  c32* ys = xs + (sync_lp(p.HX + T) + 2);
  float* ms = reinterpret_cast<float*>(ys + (sync_lp(p.HY + T) + 2));
  unsigned char* misc = smem + (((size_t)((unsigned char*)(ms + sync_lp(p.HM + T) + 2) - smem) + 15) & ~(size_t)15);
  long long* sc_i64 = reinterpret_cast<long long*>(misc);      // 3 * 5 entries
  double* sc_f64 = reinterpret_cast<double*>(misc + 128);      // 2 * 5 entries
  int* sc_i32 = reinterpret_cast<int*>(misc + 256);            // 5 entries
  unsigned char* cm = misc + 320;                               // 256 candidate masks
  unsigned long long* bc = reinterpret_cast<unsigned long long*>(misc + 576);  // 2 broadcast words

  const uint64_t seg = blockIdx.x;
  const uint64_t tile_own0 = seg * (uint64_t)p.tiles_per_seg;
  uint64_t tile_own1 = tile_own0 + (uint64_t)p.tiles_per_seg;
  if (tile_own1 > p.ntiles) tile_own1 = p.ntiles;
  const bool warm = seg > 0;
  const uint64_t tile_first = warm ? tile_own0 - (uint64_t)p.nwarm : tile_own0;
  const uint64_t ws = tile_first * (uint64_t)T;
  const uint64_t qvalid = warm ? ws + (uint64_t)p.D : 0;
  const uint64_t mvalid = warm ? ws + 2ull * (uint64_t)p.D - 1 : 0;

  // ---- segment prologue: x history from the stream, y / M history zero ---------
  for (int i = tid; i < p.HX; i += SYNC_THREADS) {
    const int64_t n = (int64_t)ws - (int64_t)p.HX + i;
    c32 v = mk(0.f, 0.f);
    if (n >= 0 && (uint64_t)n < p.nsamples) v = p.x[n];
    xs[sync_lp(i)] = v;
  }
  for (int i = tid; i < p.HY; i += SYNC_THREADS) ys[sync_lp(i)] = mk(0.f, 0.f);
  for (int i = tid; i < p.HM; i += SYNC_THREADS) ms[sync_lp(i)] = 0.0f;
  long long wpr = 0, wpi = 0, wr = 0, wm = 0;  // moving sums at the sample before the tile
  const bool x_al16 = ((uintptr_t)p.x & 15) == 0;
  const bool y_al16 = ((uintptr_t)p.y & 15) == 0;
  __syncthreads();

  for (uint64_t tile = tile_first; tile < tile_own1; tile++) {
    const uint64_t t0 = tile * (uint64_t)T;
    const bool owned = tile >= tile_own0;

    // ---- 1. load the tile of x into LDS (coalesced, 16 B per lane when aligned) ---
    if (x_al16 && t0 + (uint64_t)T <= p.nsamples) {
      const float4* src = reinterpret_cast<const float4*>(p.x + t0);
#pragma unroll
      for (int r = 0; r < SYNC_V / 2; r++) {
        const int pi = tid + r * SYNC_THREADS;  // pair index
        const float4 v = src[pi];
        const int li = sync_lp(p.HX + 2 * pi);
        xs[li] = mk(v.x, v.y);
        xs[li + 1] = mk(v.z, v.w);
      }
    } else {
#pragma unroll
      for (int r = 0; r < SYNC_V; r++) {
        const int i = tid + r * SYNC_THREADS;
        const uint64_t n = t0 + (uint64_t)i;
        c32 v = mk(0.f, 0.f);
        if (n < p.nsamples) v = p.x[n];
        xs[sync_lp(p.HX + i)] = v;
      }
    }
    __syncthreads();

    // ---- 2. channel filter: one fmaf chain per output, taps in order ----------------
    c32 acc[SYNC_V];
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) acc[j] = mk(0.f, 0.f);
    {
      const int base = p.HX + SYNC_V * tid;  // xs index of output 0 of this thread
      c32 w[15];
#pragma unroll
      for (int d = 0; d < 15; d++) w[d] = xs[sync_lp(base - 7 + d)];
      for (int kb = 0; kb < p.ntaps_pad; kb += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float hk = p.taps[kb + i];
#pragma unroll
          for (int j = 0; j < SYNC_V; j++) {
            acc[j].re = fmaf(hk, w[j - i + 7].re, acc[j].re);
            acc[j].im = fmaf(hk, w[j - i + 7].im, acc[j].im);
          }
        }
        // slide the window down by 8 samples
#pragma unroll
        for (int d = 14; d >= 8; d--) w[d] = w[d - 8];
        if (kb + 8 < p.ntaps_pad) {
#pragma unroll
          for (int d = 0; d < 8; d++) w[d] = xs[sync_lp(base - kb - 15 + d)];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) ys[sync_lp(p.HY + SYNC_V * tid + j)] = acc[j];
    __syncthreads();

    // ---- 3. y to HBM (owned tiles only), coalesced from LDS ----------------------------
    if (owned) {
      if (y_al16 && t0 + (uint64_t)T <= p.nsamples) {
        float4* dst = reinterpret_cast<float4*>(p.y + t0);
#pragma unroll
        for (int r = 0; r < SYNC_V / 2; r++) {
          const int pi = tid + r * SYNC_THREADS;
          const int li = sync_lp(p.HY + 2 * pi);
          const c32 a = ys[li], b = ys[li + 1];
          dst[pi] = make_float4(a.re, a.im, b.re, b.im);
        }
      } else {
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = tid + r * SYNC_THREADS;
          const uint64_t n = t0 + (uint64_t)i;
          if (n < p.nsamples) p.y[n] = ys[sync_lp(p.HY + i)];
        }
      }
    }

    // ---- 4. Schmidl-Cox moving sums in Q23.40 --------------------------------------------
    long long dpr[SYNC_V], dpi[SYNC_V], dr[SYNC_V];
    long long tpr = 0, tpi = 0, tr = 0;
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) {
      const int i = SYNC_V * tid + j;
      const uint64_t n = t0 + (uint64_t)i;
      const c32 a = ys[sync_lp(p.HY + i)];
      const c32 d1 = ys[sync_lp(p.HY + i - p.D)];
      const c32 d2 = ys[sync_lp(p.HY + i - 2 * p.D)];
      long long npr = 0, npi = 0, nr = 0, opr = 0, opi = 0, orr = 0;
      if (n >= qvalid) {
        const c32 c = cmul_conj(a, d1);
        npr = q40_clamped(c.re);
        npi = q40_clamped(c.im);
        nr = q40_clamped(a.re * a.re + a.im * a.im);
      }
      if (n >= qvalid + (uint64_t)p.D) {
        const c32 c = cmul_conj(d1, d2);
        opr = q40_clamped(c.re);
        opi = q40_clamped(c.im);
        orr = q40_clamped(d1.re * d1.re + d1.im * d1.im);
      }
      tpr += npr - opr;
      tpi += npi - opi;
      tr += nr - orr;
      dpr[j] = tpr;
      dpi[j] = tpi;
      dr[j] = tr;
    }
    long long totpr, totpi, totr;
    const long long epr = block_excl_scan_add<long long>(tpr, sc_i64, &totpr);
    const long long epi = block_excl_scan_add<long long>(tpi, sc_i64 + 5, &totpi);
    const long long er = block_excl_scan_add<long long>(tr, sc_i64 + 10, &totr);
    c32 Pv[SYNC_V];
    float Mv[SYNC_V];
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) {
      const uint64_t n = t0 + (uint64_t)(SYNC_V * tid + j);
      const float pre = (float)q40_to_double(wpr + epr + dpr[j]);
      const float pim = (float)q40_to_double(wpi + epi + dpi[j]);
      const float r = (float)q40_to_double(wr + er + dr[j]);
      const float num = pre * pre + pim * pim;
      const float den = r * r;
      float m = (den > 0.0f) ? (num / den) : 0.0f;
      if (!(m <= 1024.0f)) m = 1024.0f;
      if (n < mvalid) m = 0.0f;
      Pv[j] = mk(pre, pim);
      Mv[j] = m;
      ms[sync_lp(p.HM + SYNC_V * tid + j)] = m;
    }
    wpr += totpr;
    wpi += totpi;
    wr += totr;
    __syncthreads();

    // ---- 5. CP-length moving average of M, minus one -----------------------------------------
    long long dm[SYNC_V];
    long long tm = 0;
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) {
      const int i = SYNC_V * tid + j;
      const float mold = ms[sync_lp(p.HM + i - p.CP)];
      tm += q40_from_float(Mv[j]) - q40_from_float(mold);
      dm[j] = tm;
    }
    long long totm;
    const long long em = block_excl_scan_add<long long>(tm, sc_i64, &totm);
    float u[SYNC_V];
#pragma unroll
    for (int j = 0; j < SYNC_V; j++) {
      const double s = q40_to_double(wm + em + dm[j]);
      const float mbar = (float)(s * (double)p.tapcp);
      u[j] = mbar + (-1.0f);
    }
    wm += totm;

    if (owned) {
      // ---- 6. per-tile summary of the detector's running average + candidates --------------
      int nv = 0;  // valid (in-stream) samples of this thread
      unsigned cmask = 0;
      Aff f;
      f.A = 1.0;
      f.b = 0.0;
#pragma unroll
      for (int j = 0; j < SYNC_V; j++) {
        const uint64_t n = t0 + (uint64_t)(SYNC_V * tid + j);
        if (n < p.nsamples) {
          nv++;
          f.A = f.A * p.decay;
          f.b = (double)p.alpha * (double)u[j] + p.decay * f.b;
          if (u[j] > p.cand_thr) cmask |= 1u << j;
          if (p.metric_tap) p.metric_tap[n] = u[j];
        }
      }
      // inclusive scan of the affine maps across the block
      Aff inc = f;
      {
        const int lane = lane_id(), w = wave_id();
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
          Aff o;
          o.A = __shfl_up(inc.A, d, WAVE);
          o.b = __shfl_up(inc.b, d, WAVE);
          if (lane >= d) inc = aff_then(o, inc);
        }
        if (lane == WAVE - 1) {
          sc_f64[2 * w] = inc.A;
          sc_f64[2 * w + 1] = inc.b;
        }
      }
      const int anyc = __syncthreads_or(cmask != 0);
      Aff pre;  // map of everything before this thread in the tile
      pre.A = 1.0;
      pre.b = 0.0;
      Aff tot = pre;
      {
        const int w = wave_id();
        for (int i = 0; i < SYNC_THREADS / WAVE; i++) {
          Aff g;
          g.A = sc_f64[2 * i];
          g.b = sc_f64[2 * i + 1];
          if (i < w) pre = aff_then(pre, g);
          tot = aff_then(tot, g);
        }
        // exclusive within the wave: inc = pre_wave_lanes then f  =>  strip f via shuffle
        Aff prev;
        prev.A = __shfl_up(inc.A, 1, WAVE);
        prev.b = __shfl_up(inc.b, 1, WAVE);
        if (lane_id() == 0) {
          prev.A = 1.0;
          prev.b = 0.0;
        }
        pre = aff_then(pre, prev);
      }
      if (tid == 0) {
        p.tile_B[tile] = tot.b;
        if (!anyc) p.tile_npieces[tile] = 0;
      }

      if (anyc) {
        // ---- 7. candidate pieces: maximal runs of u > theta inside the tile -----------------
        cm[tid] = (unsigned char)cmask;
        __syncthreads();
        const unsigned prevbit = (tid > 0) ? ((cm[tid - 1] >> 7) & 1u) : 0u;
        const unsigned nextbit = (tid < SYNC_THREADS - 1) ? (cm[tid + 1] & 1u) : 0u;
        const unsigned ext = (cmask << 1) | prevbit;           // bit j+1 = cand[j], bit 0 = cand[-1]
        const unsigned startmask = cmask & ~ext & 0xFFu;       // cand[j] && !cand[j-1]
        const unsigned extn = (cmask >> 1) | (nextbit << 7);   // bit j = cand[j+1]
        const unsigned endmask = cmask & ~extn & 0xFFu;        // cand[j] && !cand[j+1]
        const int packed = (__popc(startmask) << 16) | __popc(cmask);
        int ptot;
        const int pex = block_excl_scan_add<int>(packed, sc_i32, &ptot);
        const int nstart_before = pex >> 16, ncand_before = pex & 0xFFFF;
        const int npieces = ptot >> 16, ncand = ptot & 0xFFFF;
        if (tid == 0) {
          unsigned long long basev = atomicAdd(p.cand_count, (unsigned long long)ncand);
          unsigned long long basep = atomicAdd(p.piece_count, (unsigned long long)npieces);
          bc[0] = basev;
          bc[1] = basep;
          if (basev + (unsigned long long)ncand > p.cand_cap || basep + (unsigned long long)npieces > p.piece_cap)
            atomicOr(p.overflow, 1u);
        }
        __syncthreads();
        const unsigned long long basev = bc[0], basep = bc[1];
        const bool fits = (basev + (unsigned long long)ncand <= p.cand_cap) &&
                          (basep + (unsigned long long)npieces <= p.piece_cap);
        if (fits) {
          double a_loc = pre.b;  // zero-init average just before this thread's first sample
          int so = nstart_before, co = ncand_before;
#pragma unroll
          for (int j = 0; j < SYNC_V; j++) {
            const uint64_t n = t0 + (uint64_t)(SYNC_V * tid + j);
            if ((startmask >> j) & 1u) {
              SyncPiece* pc = p.pieces + basep + so;
              pc->start = n;
              pc->val_off = basev + (unsigned long long)co;
              pc->bloc = a_loc;
              so++;
            }
            if ((cmask >> j) & 1u) {
              p.cand_u[basev + co] = u[j];
              p.cand_P[basev + co] = Pv[j];
              co++;
            }
            if ((endmask >> j) & 1u) p.pieces[basep + so - 1].end = n;
            if (j < nv) a_loc = (double)p.alpha * (double)u[j] + p.decay * a_loc;
          }
        }
        if (tid == 0) {
          p.tile_npieces[tile] = fits ? (uint32_t)npieces : 0u;
          p.tile_first[tile] = basep;
        }
      }
    }
    __syncthreads();

    // ---- 8. slide the histories ---------------------------------------------------------------
    if (tile + 1 < tile_own1) {
      for (int off = 0; off < p.HX; off += T) {
        c32 v[SYNC_V];
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HX && i < off + T) v[r] = xs[sync_lp(i + T)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HX && i < off + T) xs[sync_lp(i)] = v[r];
        }
        __syncthreads();
      }
      for (int off = 0; off < p.HY; off += T) {
        c32 v[SYNC_V];
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HY && i < off + T) v[r] = ys[sync_lp(i + T)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HY && i < off + T) ys[sync_lp(i)] = v[r];
        }
        __syncthreads();
      }
      for (int off = 0; off < p.HM; off += T) {
        float v[SYNC_V];
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HM && i < off + T) v[r] = ms[sync_lp(i + T)];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SYNC_V; r++) {
          const int i = off + tid + r * SYNC_THREADS;
          if (i < p.HM && i < off + T) ms[sync_lp(i)] = v[r];
        }
        __syncthreads();
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// avg carry-in per tile: avg_in[g] = running average just before the tile's first sample
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_avg_carry(const double* __restrict__ tile_B, uint64_t ntiles, uint64_t nsamples,
                                                    double decay, double* __restrict__ avg_in) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ntiles) return;
  // every earlier tile is full (only the last tile of the stream can be short)
  const double A = pow(decay, (double)SYNC_TILE);
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
  double decay;
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
};

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
    float avg = (float)(p.avg_in[g] * pow(p.decay, (double)(pc.start - g * (uint64_t)SYNC_TILE)) + pc.bloc);
    int state = 0;
    float peak_val = -INFINITY;
    uint64_t peak_ind = 0;
    c32 peak_P = mk(0.f, 0.f);
    bool open_at_stream_end = false;
    for (;;) {
      const uint64_t len = pc.end - pc.start + 1;
      for (uint64_t k = 0; k < len; k++) {
        const float u = p.cand_u[pc.val_off + k];
        const uint64_t i = pc.start + k;
        for (;;) {
          if (state == 0) {
            if (u > avg * p.rise) {
              state = 1;
              continue;
            }
            avg = p.alpha * u + one_m_alpha * avg;
            break;
          }
          if (u > peak_val) {
            peak_val = u;
            peak_ind = i;
            peak_P = p.cand_P[pc.val_off + k];
            avg = p.alpha * u + one_m_alpha * avg;
            break;
          }
          if (u > avg * p.fall) {
            avg = p.alpha * u + one_m_alpha * avg;
            break;
          }
          if (WRITE) {
            p.peaks[wbase + nflag] = peak_ind;
            p.peak_P[wbase + nflag] = peak_P;
          }
          nflag++;
          state = 0;
          peak_val = -INFINITY;
        }
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
        p.peak_P[wbase + nflag] = peak_P;
      }
      nflag++;
    }
  }
  if (!WRITE) p.counts[g0] = nflag;
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
