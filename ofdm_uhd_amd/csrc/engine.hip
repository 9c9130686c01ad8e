// engine.hip -- libofdm_hip.so: handle, workspaces and the C ABI of include/ofdm_hip.h.
// gfx950 (MI355X) only.  Build: see Makefile (hipcc --offload-arch=gfx950 -ffp-contract=off).
#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "fft.h"
#include "host_util.h"
#include "tx.h"
#include "rx_sync.h"
#include "rx_demod.h"
#include "sense.h"

static std::string g_create_error;

#define HIPCHK(h, expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
      return OFDM_E_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define FAIL(h, code, msg) \
  do {                     \
    (h)->err = (msg);      \
    return (code);         \
  } while (0)

// spectrum-sensor workspaces (sense.h / engine_sense.inc)
struct SenseState {
  DevBuf d_win, d_tw, d_msgs, d_mean, d_bits, d_hex, x_stage;
  std::vector<float> tab_win;  // window the device tables were built from
  int tab_S = 0;
  int S = 0;
  uint64_t nm = 0, nd = 0;     // messages / decisions of the last run
  bool rx_on = false;          // fused into ofdm_rx
  ofdm_sense_cfg rx_cfg;
  hipStream_t side = nullptr;  // sensing runs beside the receiver on this stream
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
};

struct ofdm_handle {
  ofdm_cfg cfg;
  int N = 0, CP = 0, L = 0, occ = 0, nbits = 0, zl = 0;
  int nc = 0;    // data carriers of the mapper (map into the N bins)
  int nmap = 0;  // data carriers of the frame sink (map into the occupied block)
  bool dev_ptrs = false;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // The transmit side runs on a stream of its own (txs) so that a caller may queue the next batch's modulation
  // behind the receiver's input stage (ofdm_rx_submit) and let it fill the receiver's host round trips:
  //   ev_tx_done  end of the last queued transmit batch      -> the receiver's stream waits for it
  //   ev_rx_in    the receiver has consumed its input buffer  -> the next transmit batch waits for it
  //   ev_tx_staged  the last batch's metadata left the pinned staging buffer
  hipStream_t own_txs = nullptr, txs = nullptr;
  hipEvent_t ev_tx_done = nullptr, ev_rx_in = nullptr, ev_tx_staged = nullptr;
  bool tx_pending = false, rx_in_pending = false, tx_staged_pending = false;
  // The receiver's reads of its input, newest and one before (a caller that alternates two IQ buffers lets batch
  // i+1 be modulated into one while batch i is still filtered out of the other): a transmit batch waits for those
  // whose buffer its output overlaps.  [0] is ev_rx_in itself.
  hipEvent_t ev_rx_in_old = nullptr;
  bool rx_in_old_pending = false;
  uintptr_t rx_in_lo[2] = {0, 0}, rx_in_hi[2] = {0, 0};
  std::string err;

  // constant tables
  DevBuf d_const, d_preamble, d_tw, d_bin2car, d_mask, d_crc, d_Hf, d_twF, d_ks, d_smap, d_kd, d_xp8, d_grid, d_synctab;
  bool has_grid = false;  // the constellation is a full grid of levels (QAM tables): constant-time slicer
  int sign_kind = 0;      // two-level constellations: sign slicer (DemodParams::sign_kind)
  unsigned char sign_idx[4] = {0, 0, 0, 0};
  float sign_eps = 0.f, sign_bound = 0.f;
  int filtF = 0;  // transform length of the channel filter (sync_filter_F)

  // TX workspaces
  DevBuf d_payloads, d_payload_off, d_payload_len, d_framed, d_framed_off, d_sym_off, d_sym_pkt, d_iq_stage,
      d_freq_tap, d_ifft_tap;
  std::vector<uint64_t> last_sym_off;  // symbol offsets of the last transmitted batch (TX_MAPPER tap)
  PinBuf h_meta;
  std::vector<uint64_t> framed_off, sym_off;
  uint64_t last_tx_nsym = 0, last_tx_framed_bytes = 0;
  int last_tx_npkt = 0;

  // channel
  bool chan_on = false;
  ofdm_chan chan;

  uint32_t tap_mask = 0;
  Profiler prof;

  RxState rx;  // receive-side workspaces (rx_demod.h)
  SenseState sense;
};

// ------------------------------------------------------------------------------
static int ilog2_ceil(unsigned v) {
  int n = 0;
  while ((1u << n) < v) n++;
  return n;
}

// digital_ofdm_mapper_bcv / digital_ofdm_frame_sink carrier map from a hex string (default "FE7F",
// transmit_path.py:64).  Both constructors grow the string with 'f' on both sides until it covers occ carriers
// (a last partial nibble split ceil(diff/2) left, the rest right); MSB of a digit = lowest carrier of its nibble.
//   mapper (sink == false): the string is centred in the fft_length bins in units of four carriers --
//       bin 4*(i + pad) + j, pad = (container/4 - digits)/2;
//   frame sink (sink == true): carrier 4*i + j - diff_left of the occupied block, diff_left = the carriers the
//       partial nibble put on the left, over the first occ/4 + diff_left digits only (its loop bound: a longer
//       string is silently clipped, a missing digit reads as 0).
// The two agree on the data carriers whenever the mapper's first carrier is bin zeros_on_left; when they do not
// (a string longer than occ/4 digits, occ % 4 != 0) the reference's own TX and RX disagree and so do these.
static int build_carrier_map(int occ, int container, const char* carriers, bool sink, std::vector<int>& map) {
  std::vector<int> digits;
  if (occ < 16) return OFDM_E_INVAL;
  if (!carriers || !carriers[0]) carriers = "FE7F";
  for (const char* c = carriers; *c; c++) {
    int v;
    if (*c >= '0' && *c <= '9') v = *c - '0';
    else if (*c >= 'a' && *c <= 'f') v = *c - 'a' + 10;
    else if (*c >= 'A' && *c <= 'F') v = *c - 'A' + 10;
    else return OFDM_E_INVAL;
    digits.push_back(v);
    if (digits.size() > OFDM_MAX_CARRIER_HEX) return OFDM_E_INVAL;
  }
  int diff = occ - 4 * (int)digits.size();
  while (diff > 7) {
    digits.insert(digits.begin(), 0xF);
    digits.push_back(0xF);
    diff -= 8;
  }
  int dl = 0;
  if (diff > 0) {
    dl = (diff + 1) / 2;
    const int dr = diff - dl;
    digits.insert(digits.begin(), (1 << dl) - 1);
    digits.push_back(0xF ^ ((1 << dr) - 1));
  }
  map.clear();
  if (sink) {
    const int nread = occ / 4 + dl;
    for (int i = 0; i < nread; i++) {
      const int d = i < (int)digits.size() ? digits[i] : 0;
      for (int j = 0; j < 4; j++)
        if ((d >> (3 - j)) & 1) {
          const int idx = 4 * i + j - dl;
          if (idx < 0 || idx >= occ) return OFDM_E_INVAL;  // (the block would read outside its input vector)
          map.push_back(idx);
        }
    }
  } else {
    const int pad = (container / 4 - (int)digits.size()) / 2;  // C integer division, as the block does
    for (size_t i = 0; i < digits.size(); i++)
      for (int j = 0; j < 4; j++)
        if ((digits[i] >> (3 - j)) & 1) {
          int idx = 4 * ((int)i + pad) + j;
          if (idx < 0 || idx >= container) return OFDM_E_INVAL;
          map.push_back(idx);
        }
  }
  if ((int)map.size() > occ) return OFDM_E_INVAL;  // "subcarriers allocated exceeds size of occupied carriers"
  if (map.empty()) return OFDM_E_INVAL;            // no data carrier at all: nothing could ever be sent
  return OFDM_OK;
}

template <typename T>
static hipError_t upload(DevBuf& b, const T* src, size_t count) {
  hipError_t e = b.ensure(sizeof(T) * std::max<size_t>(count, 1));
  if (e != hipSuccess) return e;
  if (count) return hipMemcpy(b.p, src, sizeof(T) * count, hipMemcpyHostToDevice);
  return hipSuccess;
}

static uint32_t npadding_bytes(uint32_t pkt_byte_len) {
  // _npadding_bytes(len, samples_per_symbol=1, bits_per_symbol=1) as ofdm.py:144 calls it
  uint32_t r = pkt_byte_len % 16u;
  return r == 0 ? 0 : 16u - r;
}

static int framed_len_of(const ofdm_cfg& cfg, uint32_t payload_len, uint32_t* out) {
  uint32_t Lp = payload_len + 4;
  if (Lp > OFDM_MASK_LEN) return OFDM_E_INVAL;  // "len(payload) must be in [0, 4096]" (ofdm_packet_utils.py:123-126)
  uint32_t n = 4 + Lp + 1;
  if (cfg.flags & OFDM_F_PAD_FOR_USRP) n += npadding_bytes(n);
  if (cfg.whitener_offset + (n - 4) > OFDM_MASK_LEN) return OFDM_E_INVAL;  // whitening mask exhausted
  *out = n;
  return OFDM_OK;
}

static TxParams make_tx_params(const ofdm_handle* h) {
  TxParams p;
  memset(&p, 0, sizeof(p));
  p.N = h->N;
  p.CP = h->CP;
  p.L = h->L;
  p.occ = h->occ;
  p.nc = h->nc;
  p.nbits = h->nbits;
  p.arity = (int)h->cfg.arity;
  p.zl = h->zl;
  p.scale1 = (float)(1.0 / sqrt((double)h->N));
  p.amp = h->cfg.tx_amplitude;
  p.pad_seed = h->cfg.pad_seed;
  p.whitener_offset = h->cfg.whitener_offset;
  p.pad_for_usrp = (h->cfg.flags & OFDM_F_PAD_FOR_USRP) ? 1u : 0u;
  p.constellation = h->d_const.as<c32>();
  p.preamble = h->d_preamble.as<c32>();
  p.bin2car = h->d_bin2car.as<int16_t>();
  p.tw = h->d_tw.as<c32>();
  p.mask = h->d_mask.as<uint8_t>();
  p.crc_table = h->d_crc.as<uint32_t>();
  p.chan_on = h->chan_on ? 1 : 0;
  p.sigma = h->chan.sigma;
  p.cfo = h->chan.cfo;
  p.seed = h->chan.seed;
  p.stream = h->chan.stream_id;
  return p;
}

// ------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------
extern "C" int ofdm_abi_version(void) { return OFDM_ABI_VERSION; }

extern "C" int ofdm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char* ofdm_last_error(const ofdm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" const char* ofdm_kernel_name(int k) {
  static const char* names[OFDM_K_COUNT] = {"k_frame_pack", "k_tx_mod",   "k_channel", "k_sync",
                                            "k_peak",       "k_rx_demod", "k_deframe", "k_sense",   "k_chan_filter", "k_sync_exact", "k_front"};
  return (k >= 0 && k < OFDM_K_COUNT) ? names[k] : "?";
}

// (re)build the mapper's bin->carrier table and the frame sink's carrier list
static int apply_carrier_map(ofdm_handle* h, const char* hex) {
  const int N = (int)h->cfg.fft_length, occ = (int)h->cfg.occupied_tones;
  std::vector<int> cmap, smap;
  if (build_carrier_map(occ, N, hex, false, cmap) != OFDM_OK)
    FAIL(h, OFDM_E_INVAL, "carrier map: bad hex digit, or subcarriers allocated exceeds size of occupied carriers (mapper)");
  if (build_carrier_map(occ, occ, hex, true, smap) != OFDM_OK)
    FAIL(h, OFDM_E_INVAL, "carrier map: bad hex digit, or subcarriers allocated exceeds size of occupied carriers (frame sink)");
  std::vector<int16_t> bin2car(N, (int16_t)-1);
  for (size_t i = 0; i < cmap.size(); i++) bin2car[cmap[i]] = (int16_t)i;
  std::vector<int16_t> smap16(smap.begin(), smap.end());
  // work in flight may still read the old tables
  if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->txs) HIPCHK(h, hipStreamSynchronize(h->txs));
  HIPCHK(h, upload(h->d_bin2car, bin2car.data(), bin2car.size()));
  HIPCHK(h, upload(h->d_smap, smap16.data(), smap16.size()));
  h->nc = (int)cmap.size();
  h->nmap = (int)smap.size();
  return OFDM_OK;
}

static int create_impl(const ofdm_cfg* cfg, ofdm_handle* h) {
  h->cfg = *cfg;
  const int N = (int)cfg->fft_length, occ = (int)cfg->occupied_tones, CP = (int)cfg->cp_length;
  if (N < 64 || N > OFDM_MAX_FFT || (N & (N - 1))) FAIL(h, OFDM_E_INVAL, "fft_length must be a power of two in [64, 4096]");
  if (occ > N) FAIL(h, OFDM_E_INVAL, "occupied_tones > fft_length (digital_ofdm_mapper_bcv ctor)");
  if (occ < 16) FAIL(h, OFDM_E_INVAL, "occupied_tones < 16");
  if (CP < 1 || CP > N) FAIL(h, OFDM_E_INVAL, "cp_length must be in [1, fft_length]");
  if (cfg->arity < 2 || cfg->arity > OFDM_MAX_ARITY) FAIL(h, OFDM_E_INVAL, "arity must be in [2, 256]");
  for (uint32_t i = 0; i < cfg->arity; i++)
    if (!std::isfinite(cfg->constellation[i].re) || !std::isfinite(cfg->constellation[i].im))
      FAIL(h, OFDM_E_INVAL, "constellation points must be finite");
  if (cfg->ntaps < 1 || cfg->ntaps > OFDM_MAX_TAPS) FAIL(h, OFDM_E_INVAL, "ntaps must be in [1, 512]");
  if (cfg->whitener_offset > 15) FAIL(h, OFDM_E_INVAL, "whitener_offset must be between 0 and 15, inclusive");
  // (alpha <= 0.25: the closed form of the detector's running average carries weights decay^-2048 in float64)
  if (!(cfg->peak_rise > 0.f) || !(cfg->peak_fall > 0.f) || !(cfg->peak_alpha > 0.f) || !(cfg->peak_alpha <= 0.25f))
    FAIL(h, OFDM_E_INVAL, "peak detector factors must be positive, alpha in (0, 0.25]");
  if (cfg->max_fft_shift_len > 64) FAIL(h, OFDM_E_INVAL, "max_fft_shift_len too large");
  if (cfg->sync_mode != OFDM_SYNC_PN && cfg->sync_mode != OFDM_SYNC_FIXED)
    FAIL(h, OFDM_E_INVAL, "sync_mode must be OFDM_SYNC_PN or OFDM_SYNC_FIXED (\"ml\" / \"pnac\" need blocks the reference does not ship)");
  if (cfg->sync_mode == OFDM_SYNC_FIXED && cfg->fixed_nsymbols < 1) FAIL(h, OFDM_E_INVAL, "fixed_nsymbols must be >= 1");
  h->N = N;
  h->CP = CP;
  h->L = N + CP;
  h->occ = occ;
  h->nbits = ilog2_ceil(cfg->arity);
  h->zl = (N - occ + 1) / 2;  // ceil((N-occ)/2), ofdm.py:71
  h->dev_ptrs = (cfg->flags & OFDM_F_DEVICE_PTRS) != 0;

  int ndev = 0;
  HIPCHK(h, hipGetDeviceCount(&ndev));
  if (cfg->device_id < 0 || cfg->device_id >= ndev) FAIL(h, OFDM_E_INVAL, "device_id out of range");
  HIPCHK(h, hipSetDevice(cfg->device_id));
  HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  HIPCHK(h, hipStreamCreateWithFlags(&h->own_txs, hipStreamNonBlocking));
  h->txs = h->own_txs;
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_tx_done, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_rx_in, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_rx_in_old, hipEventDisableTiming));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_tx_staged, hipEventDisableTiming));

  h->cfg.carrier_map[OFDM_MAX_CARRIER_HEX + 7] = 0;
  {
    int rc = apply_carrier_map(h, h->cfg.carrier_map);
    if (rc != OFDM_OK) return rc;
  }

  // tables
  std::vector<c32> pre(N), tw(N);
  for (int i = 0; i < N; i++) pre[i] = c32{0.f, 0.f};
  for (int i = 0; i < occ; i++) {
    if (h->zl + i < N) pre[h->zl + i] = c32{cfg->known_symbol[i].re, cfg->known_symbol[i].im};
  }
  for (int k = 0; k < N; k++) {
    double a = -2.0 * M_PI * (double)k / (double)N;
    tw[k] = c32{(float)cos(a), (float)sin(a)};
  }
  uint32_t crc[256];
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    crc[i] = c;
  }
  // channel filter in the frequency domain, the way gr_fft_filter_ccc holds its taps: the F-point transform of
  // the zero-padded taps, scaled by 1/F (the inverse transform is unnormalised).  Computed in float64 and rounded
  // once; the oracle computes the same sums in the same order.
  h->filtF = sync_filter_F((int)cfg->ntaps);
  std::vector<c32> Hf(h->filtF), twF(h->filtF);
  {
    const int F = h->filtF;
    std::vector<double> cs(F), sn(F);
    for (int m = 0; m < F; m++) {
      const double a = -2.0 * M_PI * (double)m / (double)F;
      cs[m] = cos(a);
      sn[m] = sin(a);
      twF[m] = c32{(float)cs[m], (float)sn[m]};
    }
    for (int k = 0; k < F; k++) {
      double re = 0.0, im = 0.0;
      for (int n = 0; n < (int)cfg->ntaps; n++) {
        const int idx = (int)(((long long)k * n) % F);
        re = re + (double)cfg->taps[n] * cs[idx];
        im = im + (double)cfg->taps[n] * sn[idx];
      }
      Hf[k] = c32{(float)(re / (double)F), (float)(im / (double)F)};
    }
  }
  // known_phase_diff of digital_ofdm_frame_acquisition's ctor
  std::vector<float> kd(occ, 0.0f);
  for (int i = 0; i + 2 < occ; i += 2) {
    float dr = cfg->known_symbol[i].re - cfg->known_symbol[i + 2].re;
    float di = cfg->known_symbol[i].im - cfg->known_symbol[i + 2].im;
    kd[i] = dr * dr + di * di;
  }
  HIPCHK(h, upload(h->d_const, reinterpret_cast<const c32*>(cfg->constellation), cfg->arity));
  {
    // grid structure of the constellation (qam.py:29-73 builds every combination of its levels): distinct real and
    // imaginary parts by exact float comparison, every pair present once
    SlicerGrid g;
    memset(&g, 0, sizeof(g));
    std::vector<float> lr, li;
    for (uint32_t i = 0; i < cfg->arity; i++) {
      lr.push_back(cfg->constellation[i].re);
      li.push_back(cfg->constellation[i].im);
    }
    std::sort(lr.begin(), lr.end());
    std::sort(li.begin(), li.end());
    lr.erase(std::unique(lr.begin(), lr.end()), lr.end());
    li.erase(std::unique(li.begin(), li.end()), li.end());
    bool ok = cfg->arity >= 16 && lr.size() >= 2 && li.size() >= 2 && lr.size() <= 16 && li.size() <= 16 &&
              lr.size() * li.size() == cfg->arity;
    if (ok) {
      std::vector<int> seen(cfg->arity, 0);
      float amax = 0.f;
      for (uint32_t i = 0; i < cfg->arity && ok; i++) {
        const float re = cfg->constellation[i].re, im = cfg->constellation[i].im;
        const size_t a = std::lower_bound(lr.begin(), lr.end(), re) - lr.begin();
        const size_t b = std::lower_bound(li.begin(), li.end(), im) - li.begin();
        const size_t cell = a * li.size() + b;
        if (cell >= seen.size() || seen[cell]++) {
          ok = false;
          break;
        }
        g.idx[cell] = (uint8_t)i;
        amax = fmaxf(amax, fmaxf(fabsf(re), fabsf(im)));
      }
      // The four bracketing points hold the full search's first minimum only while a step between neighbouring levels
      // is not absorbed by the rounding of |x - pos|^2 at the edge of the slicer's range (bound = 64 amax, beyond
      // which the full search runs): smallest spacing^2 above the float32 ulp of 2 bound^2.
      if (ok) {
        float dmin = INFINITY;
        for (size_t a = 1; a < lr.size(); a++) dmin = fminf(dmin, lr[a] - lr[a - 1]);
        for (size_t b = 1; b < li.size(); b++) dmin = fminf(dmin, li[b] - li[b - 1]);
        const float bnd = 64.0f * amax;
        if (!(dmin * dmin > 2.0f * bnd * bnd * 1.1920929e-7f)) ok = false;
      }
      g.nr = (int)lr.size();
      g.ni = (int)li.size();
      g.bound = 64.0f * amax;
      for (size_t a = 0; a < lr.size(); a++) g.lr[a] = lr[a];
      for (size_t b = 0; b < li.size(); b++) g.li[b] = li[b];
    }
    h->has_grid = ok;
    if (ok) HIPCHK(h, upload(h->d_grid, &g, (size_t)1));
  }
  {
    // two-level constellations (psk.py:27-60 bpsk / qpsk with the rotation of ofdm.py:94-101): sign slicer
    const ofdm_c32* c = cfg->constellation;
    h->sign_kind = 0;
    if (cfg->arity == 2 && c[0].im == 0.f && c[1].im == 0.f && c[0].re == -c[1].re && c[0].re != 0.f) {
      h->sign_kind = 1;
      h->sign_idx[c[0].re > 0.f ? 1 : 0] = 0;
      h->sign_idx[c[1].re > 0.f ? 1 : 0] = 1;
    } else if (cfg->arity == 4) {
      const float a = fabsf(c[0].re), b = fabsf(c[0].im);
      bool ok4 = a > 0.f && b > 0.f;
      int seen = 0;
      for (int i = 0; i < 4 && ok4; i++) {
        if (fabsf(c[i].re) != a || fabsf(c[i].im) != b) ok4 = false;
        const int ix = (c[i].re > 0.f ? 2 : 0) | (c[i].im > 0.f ? 1 : 0);
        if (seen & (1 << ix)) ok4 = false;
        seen |= 1 << ix;
        h->sign_idx[ix] = (unsigned char)i;
      }
      if (ok4) h->sign_kind = 2;
    }
    if (h->sign_kind) {
      // margins: the distances of two table entries differ by 4 |level| |part| >= 4 level eps, against a rounding of
      // about 4 ulp of distances below 2 (bound + level)^2: eps = level / 512, bound = 8 levels -> factor > 200
      const float lvl = h->sign_kind == 1 ? fabsf(c[0].re) : fminf(fabsf(c[0].re), fabsf(c[0].im));
      h->sign_eps = lvl * (1.0f / 512.0f);
      h->sign_bound = 8.0f * fmaxf(fabsf(c[0].re), fabsf(c[0].im));
    }
  }
  HIPCHK(h, upload(h->d_preamble, pre.data(), pre.size()));
  HIPCHK(h, upload(h->d_tw, tw.data(), tw.size()));
  HIPCHK(h, upload(h->d_mask, cfg->whitening_mask, (size_t)OFDM_MASK_LEN));
  HIPCHK(h, upload(h->d_crc, crc, (size_t)256));
  {
    // x^(8k) mod P for k = 0..4096 in zlib's reflected representation (x^0 = 0x80000000): the operator
    // of crc32_combine for k following bytes
    std::vector<uint32_t> xp8(OFDM_MASK_LEN + 1);
    uint32_t v = 0x80000000u;
    for (size_t k = 0; k < xp8.size(); k++) {
      xp8[k] = v;
      for (int b = 0; b < 8; b++) v = (v & 1u) ? ((v >> 1) ^ 0xEDB88320u) : (v >> 1);  // times x
    }
    HIPCHK(h, upload(h->d_xp8, xp8.data(), xp8.size()));
  }
  {
    // tables of the peak detector's normative average (rx_sync.h SyncParams; the oracle builds the same ones):
    // dpow[j] = decay^j by repeated multiplication | ipow[j] = 1 / dpow[j] | wtab[t] (float)
    const double decay = (double)(1.0f - cfg->peak_alpha);
    std::vector<double> tab(2 * (SYNC_TILE + 1) + SYNC_THREADS / 2);
    double* dp = tab.data();
    double* ip = dp + (SYNC_TILE + 1);
    float* wt = reinterpret_cast<float*>(ip + (SYNC_TILE + 1));
    dp[0] = 1.0;
    for (int j = 1; j <= SYNC_TILE; j++) dp[j] = dp[j - 1] * decay;
    for (int j = 0; j <= SYNC_TILE; j++) ip[j] = 1.0 / dp[j];
    for (int t = 0; t < SYNC_THREADS; t++) wt[t] = (float)dp[SYNC_TILE - SYNC_V * (t + 1)];
    HIPCHK(h, upload(h->d_synctab, tab.data(), tab.size()));
  }
  HIPCHK(h, upload(h->d_Hf, Hf.data(), Hf.size()));
  HIPCHK(h, upload(h->d_twF, twF.data(), twF.size()));
  HIPCHK(h, upload(h->d_ks, reinterpret_cast<const c32*>(cfg->known_symbol), (size_t)occ));
  HIPCHK(h, upload(h->d_kd, kd.data(), kd.size()));
  return OFDM_OK;
}

extern "C" int ofdm_create(const ofdm_cfg* cfg, ofdm_handle** out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return OFDM_E_INVAL;
  }
  if (cfg->struct_size != sizeof(ofdm_cfg)) {
    g_create_error = "ofdm_cfg.struct_size does not match this library (ABI mismatch)";
    return OFDM_E_INVAL;
  }
  ofdm_handle* h = new (std::nothrow) ofdm_handle();
  if (!h) {
    g_create_error = "out of memory";
    return OFDM_E_NOMEM;
  }
  int rc = create_impl(cfg, h);
  if (rc != OFDM_OK) {
    g_create_error = h->err;
    ofdm_destroy(h);
    return rc;
  }
  *out = h;
  return OFDM_OK;
}

extern "C" void ofdm_destroy(ofdm_handle* h) {
  if (!h) return;
  if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
  if (h->own_txs) (void)hipStreamSynchronize(h->own_txs);
  DevBuf* bufs[] = {&h->d_const,    &h->d_preamble,    &h->d_tw,          &h->d_bin2car, &h->d_mask,
                    &h->d_crc,      &h->d_Hf,   &h->d_twF,     &h->d_grid,  &h->d_synctab,    &h->d_ks,          &h->d_smap,    &h->d_kd,  &h->d_xp8,
                    &h->d_payloads, &h->d_payload_off, &h->d_payload_len, &h->d_framed,  &h->d_framed_off,
                    &h->d_sym_off,  &h->d_sym_pkt,     &h->d_iq_stage,    &h->d_freq_tap, &h->d_ifft_tap};
  for (DevBuf* b : bufs) b->release();
  h->h_meta.release();
  h->rx.release();
  {
    SenseState& ss = h->sense;
    if (ss.side) (void)hipStreamSynchronize(ss.side);
    DevBuf* sb[] = {&ss.d_win, &ss.d_tw, &ss.d_msgs, &ss.d_mean, &ss.d_bits, &ss.d_hex, &ss.x_stage};
    for (DevBuf* b : sb) b->release();
    if (ss.ev_in) (void)hipEventDestroy(ss.ev_in);
    if (ss.ev_out) (void)hipEventDestroy(ss.ev_out);
    if (ss.side) (void)hipStreamDestroy(ss.side);
  }
  h->prof.destroy();
  if (h->ev_tx_done) (void)hipEventDestroy(h->ev_tx_done);
  if (h->ev_rx_in) (void)hipEventDestroy(h->ev_rx_in);
  if (h->ev_rx_in_old) (void)hipEventDestroy(h->ev_rx_in_old);
  if (h->ev_tx_staged) (void)hipEventDestroy(h->ev_tx_staged);
  if (h->own_txs) (void)hipStreamDestroy(h->own_txs);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

extern "C" int ofdm_set_stream(ofdm_handle* h, void* s) {
  if (!h) return OFDM_E_INVAL;
  // a caller-owned stream carries both sides (everything in the caller's order); NULL: the handle's own two
  h->stream = s ? (hipStream_t)s : h->own_stream;
  h->txs = s ? (hipStream_t)s : h->own_txs;
  return OFDM_OK;
}

extern "C" int ofdm_set_carrier_map(ofdm_handle* h, const char* hex) {
  if (!h) return OFDM_E_INVAL;
  if (hex && strlen(hex) > OFDM_MAX_CARRIER_HEX) FAIL(h, OFDM_E_INVAL, "carrier map longer than 1024 hex digits");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  int rc = apply_carrier_map(h, hex ? hex : "");
  if (rc != OFDM_OK) return rc;
  memset(h->cfg.carrier_map, 0, sizeof(h->cfg.carrier_map));
  if (hex) strcpy(h->cfg.carrier_map, hex);
  return OFDM_OK;
}

extern "C" int ofdm_set_tx_amplitude(ofdm_handle* h, float ampl) {
  if (!h) return OFDM_E_INVAL;
  h->cfg.tx_amplitude = fmaxf(0.0f, fminf(ampl, 1.0f));  // transmit_path.py:56-62
  return OFDM_OK;
}

extern "C" int ofdm_set_channel(ofdm_handle* h, const ofdm_chan* c) {
  if (!h) return OFDM_E_INVAL;
  if (c) {
    h->chan = *c;
    h->chan_on = true;
  } else {
    memset(&h->chan, 0, sizeof(h->chan));
    h->chan_on = false;
  }
  return OFDM_OK;
}

extern "C" int ofdm_set_taps(ofdm_handle* h, uint32_t mask) {
  if (!h) return OFDM_E_INVAL;
  h->tap_mask = mask;
  return OFDM_OK;
}

extern "C" int ofdm_prof_enable(ofdm_handle* h, int on) {
  if (!h) return OFDM_E_INVAL;
  h->prof.on = on != 0;
  return OFDM_OK;
}
extern "C" int ofdm_prof_reset(ofdm_handle* h) {
  if (!h) return OFDM_E_INVAL;
  h->prof.reset();
  return OFDM_OK;
}
extern "C" int ofdm_prof_get(ofdm_handle* h, int k, double* total_ms, uint64_t* launches) {
  if (!h || k < 0 || k >= OFDM_K_COUNT) return OFDM_E_INVAL;
  if (total_ms) *total_ms = h->prof.total_ms[k];
  if (launches) *launches = h->prof.launches[k];
  return OFDM_OK;
}

// ------------------------------------------------------------------------------
// framing + TX
// ------------------------------------------------------------------------------
extern "C" int ofdm_framed_len(const ofdm_handle* h, uint32_t payload_len, uint32_t* framed_len) {
  if (!h || !framed_len) return OFDM_E_INVAL;
  return framed_len_of(h->cfg, payload_len, framed_len);
}

// layout of a batch: framed byte offsets and symbol offsets (host)
static int plan_batch(ofdm_handle* h, const uint32_t* payload_len, int npkt, bool* uniform, uint32_t* spp) {
  h->framed_off.assign((size_t)npkt + 1, 0);
  h->sym_off.assign((size_t)npkt + 1, 0);
  const uint64_t per = (uint64_t)h->nc * (uint64_t)h->nbits;
  bool uni = true;
  uint32_t first = 0;
  for (int k = 0; k < npkt; k++) {
    uint32_t fl;
    int rc = framed_len_of(h->cfg, payload_len[k], &fl);
    if (rc) FAIL(h, rc, "len(payload) must be in [0, 4091] (payload + CRC must fit the 12-bit length and the whitening mask)");
    h->framed_off[k + 1] = h->framed_off[k] + fl;
    // a symbol is started while message bytes remain: ceil(8*len / (carriers*nbits)), plus the preamble
    uint32_t ns = (uint32_t)((8ull * fl + per - 1) / per) + 1;
    h->sym_off[k + 1] = h->sym_off[k] + ns;
    if (k == 0)
      first = ns;
    else if (ns != first)
      uni = false;
  }
  *uniform = uni && npkt > 0;
  *spp = first;
  return OFDM_OK;
}

extern "C" int ofdm_tx_frame_count(const ofdm_handle* hc, const uint32_t* payload_len, int npkt, uint64_t* nsymbols,
                                   uint64_t* nsamples) {
  ofdm_handle* h = const_cast<ofdm_handle*>(hc);
  if (!h || npkt < 0 || (npkt && !payload_len)) return OFDM_E_INVAL;
  bool uni;
  uint32_t spp;
  int rc = plan_batch(h, payload_len, npkt, &uni, &spp);
  if (rc) return rc;
  uint64_t ns = h->sym_off[npkt];
  if (nsymbols) *nsymbols = ns;
  if (nsamples) *nsamples = ns * (uint64_t)h->L + (h->chan_on ? h->chan.lead_samples + h->chan.tail_samples : 0);
  return OFDM_OK;
}

// uploads the batch metadata and (in host-pointer mode) the payload bytes
static int stage_batch(ofdm_handle* h, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* payload_len,
                       int npkt, const uint8_t** d_payloads) {
  const size_t n1 = (size_t)npkt + 1;
  // pinned staging: payload_off[npkt] | framed_off[n1] | sym_off[n1] | payload_len[npkt]
  size_t bytes = sizeof(uint64_t) * (npkt + 2 * n1) + sizeof(uint32_t) * npkt;
  if (h->tx_staged_pending) {  // the previous batch's copies out of the pinned staging buffer must be done
    HIPCHK(h, hipEventSynchronize(h->ev_tx_staged));
    h->tx_staged_pending = false;
  }
  HIPCHK(h, h->h_meta.ensure(bytes));
  uint64_t* m_poff = h->h_meta.as<uint64_t>();
  uint64_t* m_foff = m_poff + npkt;
  uint64_t* m_soff = m_foff + n1;
  uint32_t* m_plen = reinterpret_cast<uint32_t*>(m_soff + n1);
  memcpy(m_poff, payload_off, sizeof(uint64_t) * npkt);
  memcpy(m_foff, h->framed_off.data(), sizeof(uint64_t) * n1);
  memcpy(m_soff, h->sym_off.data(), sizeof(uint64_t) * n1);
  memcpy(m_plen, payload_len, sizeof(uint32_t) * npkt);
  HIPCHK(h, h->d_payload_off.ensure(sizeof(uint64_t) * n1));
  HIPCHK(h, h->d_framed_off.ensure(sizeof(uint64_t) * n1));
  HIPCHK(h, h->d_sym_off.ensure(sizeof(uint64_t) * n1));
  HIPCHK(h, h->d_payload_len.ensure(sizeof(uint32_t) * n1));
  HIPCHK(h, hipMemcpyAsync(h->d_payload_off.p, m_poff, sizeof(uint64_t) * npkt, hipMemcpyHostToDevice, h->txs));
  HIPCHK(h, hipMemcpyAsync(h->d_framed_off.p, m_foff, sizeof(uint64_t) * n1, hipMemcpyHostToDevice, h->txs));
  HIPCHK(h, hipMemcpyAsync(h->d_sym_off.p, m_soff, sizeof(uint64_t) * n1, hipMemcpyHostToDevice, h->txs));
  HIPCHK(h, hipMemcpyAsync(h->d_payload_len.p, m_plen, sizeof(uint32_t) * npkt, hipMemcpyHostToDevice, h->txs));
  HIPCHK(h, hipEventRecord(h->ev_tx_staged, h->txs));
  h->tx_staged_pending = true;
  if (h->dev_ptrs) {
    *d_payloads = payloads;
  } else {
    uint64_t total = 0;
    for (int k = 0; k < npkt; k++) total = std::max<uint64_t>(total, payload_off[k] + payload_len[k]);
    HIPCHK(h, h->d_payloads.ensure(std::max<uint64_t>(total, 1)));
    if (total) HIPCHK(h, hipMemcpyAsync(h->d_payloads.p, payloads, total, hipMemcpyHostToDevice, h->txs));
    *d_payloads = h->d_payloads.as<uint8_t>();
  }
  return OFDM_OK;
}

static int launch_frame_pack(ofdm_handle* h, const uint8_t* d_payloads, int npkt, uint8_t* d_framed) {
  TxParams p = make_tx_params(h);
  h->prof.begin(OFDM_K_FRAME, h->txs);
  hipLaunchKernelGGL(k_frame_pack, dim3((npkt + 3) / 4), dim3(256), 0, h->txs, p, d_payloads,
                     h->d_payload_off.as<uint64_t>(), h->d_payload_len.as<uint32_t>(), h->d_framed_off.as<uint64_t>(),
                     npkt, d_framed, h->d_xp8.as<uint32_t>());
  h->prof.end(h->txs);
  HIPCHK(h, hipGetLastError());
  return OFDM_OK;
}

extern "C" int ofdm_make_packets(ofdm_handle* h, const uint8_t* payloads, const uint64_t* payload_off,
                                 const uint32_t* payload_len, int npkt, uint8_t* framed, uint64_t framed_cap,
                                 uint64_t* framed_off) {
  if (!h) return OFDM_E_INVAL;
  if (npkt < 0 || (npkt && (!payloads || !payload_off || !payload_len)) || !framed_off) FAIL(h, OFDM_E_INVAL, "null argument");
  bool uni;
  uint32_t spp;
  int rc = plan_batch(h, payload_len, npkt, &uni, &spp);
  if (rc) return rc;
  memcpy(framed_off, h->framed_off.data(), sizeof(uint64_t) * ((size_t)npkt + 1));
  const uint64_t total = h->framed_off[npkt];
  if (total > framed_cap) FAIL(h, OFDM_E_CAPACITY, "framed buffer too small");
  if (npkt == 0) return OFDM_OK;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const uint8_t* d_payloads = nullptr;
  rc = stage_batch(h, payloads, payload_off, payload_len, npkt, &d_payloads);
  if (rc) return rc;
  uint8_t* d_framed = framed;
  if (!h->dev_ptrs) {
    HIPCHK(h, h->d_framed.ensure(total));
    d_framed = h->d_framed.as<uint8_t>();
  }
  rc = launch_frame_pack(h, d_payloads, npkt, d_framed);
  if (rc) return rc;
  if (!h->dev_ptrs) HIPCHK(h, hipMemcpyAsync(framed, d_framed, total, hipMemcpyDeviceToHost, h->txs));
  HIPCHK(h, hipStreamSynchronize(h->txs));
  h->prof.collect();
  return OFDM_OK;
}

template <int N>
static void launch_tx_mod(ofdm_handle* h, const TxParams& p, const uint8_t* d_framed, uint32_t uniform_spp, uint64_t nsym,
                          uint64_t lead, c32* d_out, c32* d_freq_tap, c32* d_ifft_tap) {
  constexpr int SPW = TxGeom<N>::SPW, WG = TxGeom<N>::WG;
  const size_t shmem = (size_t)TxGeom<N>::lds_bytes();  // transforms' buffers | constellation
  const unsigned grid = (unsigned)((nsym + SPW - 1) / SPW);
  // (a batch without transmit-side taps and without a carrier offset runs the kernel compiled without them: tx.h)
  if (!d_freq_tap && !d_ifft_tap && !(p.chan_on && p.cfo != 0.0f))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tx_mod<N, true>), dim3(grid), dim3(WG), shmem, h->txs, p, d_framed,
                       h->d_framed_off.as<uint64_t>(), h->d_sym_off.as<uint64_t>(), h->d_sym_pkt.as<uint32_t>(),
                       uniform_spp, nsym, lead, d_out, d_freq_tap, d_ifft_tap);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tx_mod<N, false>), dim3(grid), dim3(WG), shmem, h->txs, p, d_framed,
                       h->d_framed_off.as<uint64_t>(), h->d_sym_off.as<uint64_t>(), h->d_sym_pkt.as<uint32_t>(),
                       uniform_spp, nsym, lead, d_out, d_freq_tap, d_ifft_tap);
}

static int launch_noise(ofdm_handle* h, hipStream_t st, c32* d_iq, uint64_t n, uint64_t index0, int zero_input, const ofdm_chan& ch) {
  if (n == 0) return OFDM_OK;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 8);
  h->prof.begin(OFDM_K_CHAN, st);
  hipLaunchKernelGGL(k_channel, dim3(grid), dim3(256), 0, st, d_iq, n, index0, zero_input, ch.sigma, ch.cfo,
                     ch.seed, ch.stream_id);
  h->prof.end(st);
  HIPCHK(h, hipGetLastError());
  return OFDM_OK;
}

// everything of ofdm_tx up to, not including, the final synchronisation
static int tx_enqueue(ofdm_handle* h, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* payload_len,
                      int npkt, ofdm_c32* iq_out, uint64_t iq_cap, uint64_t* nsamples, ofdm_stats* stats) {
  if (!h) return OFDM_E_INVAL;
  if (npkt < 0 || (npkt && (!payloads || !payload_off || !payload_len)) || !nsamples) FAIL(h, OFDM_E_INVAL, "null argument");
  bool uni;
  uint32_t spp;
  int rc = plan_batch(h, payload_len, npkt, &uni, &spp);
  if (rc) return rc;
  const uint64_t nsym = h->sym_off[npkt];
  const uint64_t lead = h->chan_on ? h->chan.lead_samples : 0, tail = h->chan_on ? h->chan.tail_samples : 0;
  const uint64_t total = lead + nsym * (uint64_t)h->L + tail;
  *nsamples = total;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->symbols = nsym;
    stats->samples = total;
    stats->packets = (uint64_t)npkt;
  }
  if (total > iq_cap) FAIL(h, OFDM_E_CAPACITY, "iq_out too small (see ofdm_tx_frame_count)");
  if (total == 0) return OFDM_OK;
  if (!iq_out) FAIL(h, OFDM_E_INVAL, "null iq_out");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  // A submitted receive stage that reads its input to the end of the call (SYNC "fixed", fused sensing) has recorded
  // no event this batch could wait on: refilling THAT buffer now would race with the receiver.
  if (h->rx.sub_hold && h->rx.sub_valid) {
    const uintptr_t a0 = (uintptr_t)iq_out, a1 = a0 + total * sizeof(c32);
    const uintptr_t b0 = (uintptr_t)h->rx.sub_iq, b1 = b0 + h->rx.sub_n * sizeof(c32);
    if (a0 < b1 && b0 < a1)
      FAIL(h, OFDM_E_INVAL, "ofdm_tx into the buffer of a submitted ofdm_rx that reads it to the end of the call (SYNC fixed / fused sensing): call ofdm_rx first");
  }
  // the receiver may still be reading the buffer this batch writes (ofdm_rx_submit / ofdm_rx in flight)
  if (h->txs != h->stream) {
    const uintptr_t o0 = (uintptr_t)iq_out, o1 = o0 + total * sizeof(c32);
    // (host-pointer mode stages through the handle's own buffers: always ordered)
    const bool all = !h->dev_ptrs;
    if (h->rx_in_pending && (all || (o0 < h->rx_in_hi[0] && h->rx_in_lo[0] < o1))) HIPCHK(h, hipStreamWaitEvent(h->txs, h->ev_rx_in, 0));
    if (h->rx_in_old_pending && (all || (o0 < h->rx_in_hi[1] && h->rx_in_lo[1] < o1)))
      HIPCHK(h, hipStreamWaitEvent(h->txs, h->ev_rx_in_old, 0));
  }

  c32* d_out = reinterpret_cast<c32*>(iq_out);
  if (!h->dev_ptrs) {
    HIPCHK(h, h->d_iq_stage.ensure(total * sizeof(c32)));
    d_out = h->d_iq_stage.as<c32>();
  }
  if (npkt > 0) {
    const uint8_t* d_payloads = nullptr;
    rc = stage_batch(h, payloads, payload_off, payload_len, npkt, &d_payloads);
    if (rc) return rc;
    HIPCHK(h, h->d_framed.ensure(h->framed_off[npkt]));
    rc = launch_frame_pack(h, d_payloads, npkt, h->d_framed.as<uint8_t>());
    if (rc) return rc;
    HIPCHK(h, h->d_sym_pkt.ensure(sizeof(uint32_t) * std::max<uint64_t>(nsym, 1)));
    if (!uni) {
      hipLaunchKernelGGL(k_sym_desc, dim3((npkt + 255) / 256), dim3(256), 0, h->txs, h->d_sym_off.as<uint64_t>(), npkt,
                         h->d_sym_pkt.as<uint32_t>());
      HIPCHK(h, hipGetLastError());
    }
    c32 *d_freq = nullptr, *d_ifft = nullptr;
    if (h->tap_mask & ((1u << OFDM_TAP_TX_FREQ) | (1u << OFDM_TAP_TX_MAPPER))) {
      HIPCHK(h, h->d_freq_tap.ensure(nsym * (uint64_t)h->N * sizeof(c32)));
      d_freq = h->d_freq_tap.as<c32>();
    }
    if (h->tap_mask & (1u << OFDM_TAP_TX_IFFT)) {
      HIPCHK(h, h->d_ifft_tap.ensure(nsym * (uint64_t)h->N * sizeof(c32)));
      d_ifft = h->d_ifft_tap.as<c32>();
    }
    TxParams p = make_tx_params(h);
    const uint32_t uspp = uni ? spp : 0;
    h->prof.begin(OFDM_K_TX, h->txs);
    switch (h->N) {
      case 64: launch_tx_mod<64>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      case 128: launch_tx_mod<128>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      case 256: launch_tx_mod<256>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      case 512: launch_tx_mod<512>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      case 1024: launch_tx_mod<1024>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      case 2048: launch_tx_mod<2048>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
      default: launch_tx_mod<4096>(h, p, h->d_framed.as<uint8_t>(), uspp, nsym, lead, d_out, d_freq, d_ifft); break;
    }
    h->prof.end(h->txs);
    HIPCHK(h, hipGetLastError());
  }
  h->last_tx_nsym = nsym;
  h->last_sym_off = h->sym_off;
  h->last_tx_npkt = npkt;
  h->last_tx_framed_bytes = npkt ? h->framed_off[npkt] : 0;
  // noise-only lead-in and tail (the modulator covers everything in between)
  if (h->chan_on) {
    rc = launch_noise(h, h->txs, d_out, lead, 0, 1, h->chan);
    if (rc) return rc;
    rc = launch_noise(h, h->txs, d_out + lead + nsym * (uint64_t)h->L, tail, lead + nsym * (uint64_t)h->L, 1, h->chan);
    if (rc) return rc;
  }
  if (!h->dev_ptrs) HIPCHK(h, hipMemcpyAsync(iq_out, d_out, total * sizeof(c32), hipMemcpyDeviceToHost, h->txs));
  HIPCHK(h, hipEventRecord(h->ev_tx_done, h->txs));
  h->tx_pending = true;
  return OFDM_OK;
}

extern "C" int ofdm_wait(ofdm_handle* h) {
  if (!h) return OFDM_E_INVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipStreamSynchronize(h->txs));
  if (h->stream != h->txs) HIPCHK(h, hipStreamSynchronize(h->stream));
  h->tx_pending = false;
  h->prof.collect();
  return OFDM_OK;
}

extern "C" int ofdm_tx(ofdm_handle* h, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* payload_len,
                       int npkt, ofdm_c32* iq_out, uint64_t iq_cap, uint64_t* nsamples, ofdm_stats* stats) {
  int rc = tx_enqueue(h, payloads, payload_off, payload_len, npkt, iq_out, iq_cap, nsamples, stats);
  if (rc != OFDM_OK) return rc;
  return ofdm_wait(h);
}

extern "C" int ofdm_tx_async(ofdm_handle* h, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* payload_len,
                             int npkt, ofdm_c32* iq_out, uint64_t iq_cap, uint64_t* nsamples, ofdm_stats* stats) {
  return tx_enqueue(h, payloads, payload_off, payload_len, npkt, iq_out, iq_cap, nsamples, stats);
}

extern "C" int ofdm_channel(ofdm_handle* h, ofdm_c32* iq, uint64_t n, const ofdm_chan* chan, uint64_t index0) {
  if (!h) return OFDM_E_INVAL;
  if (!chan || (n && !iq)) FAIL(h, OFDM_E_INVAL, "null argument");
  if (n == 0) return OFDM_OK;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  c32* d = reinterpret_cast<c32*>(iq);
  // a transmit batch still in flight (ofdm_tx_async) may be using the staging buffer, or writing the caller's
  if (h->tx_pending && h->txs != h->stream) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_tx_done, 0));
  if (!h->dev_ptrs) {
    if (h->tx_pending && h->txs != h->stream) HIPCHK(h, hipStreamSynchronize(h->txs));  // (ensure() may reallocate it)
    HIPCHK(h, h->d_iq_stage.ensure(n * sizeof(c32)));
    d = h->d_iq_stage.as<c32>();
    HIPCHK(h, hipMemcpyAsync(d, iq, n * sizeof(c32), hipMemcpyHostToDevice, h->stream));
  }
  int rc = launch_noise(h, h->stream, d, n, index0, 0, *chan);
  if (rc) return rc;
  if (!h->dev_ptrs) HIPCHK(h, hipMemcpyAsync(iq, d, n * sizeof(c32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->prof.collect();
  return OFDM_OK;
}

#include "engine_sense.inc"
#include "engine_rx.inc"
