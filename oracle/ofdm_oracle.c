/*
 * ofdm_oracle.c -- CPU restatement of the rubiruchi/ofdm_uhd ofdm_mod/ofdm_demod
 * hot path.  TEST INFRASTRUCTURE ONLY: it is the parity checker for
 * libofdm_hip.so and the "port" CPU baseline of bench.py.  Nothing in the
 * product (ofdm_uhd_amd/) may import, link or call it.
 *
 * PARITY UNPINNED at the GNU Radio boundary: the arithmetic of this path lives
 * in GNU Radio 3.6.0 C++ blocks (gnuradio-core + gr-digital, pinned only by
 * path comments ofdm.py:29 and the banner output.txt:1) that are neither under
 * /root/reference nor installed, and the reference has no tests or golden
 * vectors for them.  Each function below restates the published behaviour of
 * the block named at the reference call site it cites.
 * What IS pinned, by tests/golden/reference_constants.json: the constellation
 * tables (psk.py:27-60, qam.py:29-73), the preamble sequence (ofdm.py:310-325),
 * the whitening mask (ofdm_packet_utils.py:195-452), the header layout
 * (ofdm_packet_utils.py:93-97) and CRC-32's standard check value.
 *
 * Arithmetic conventions (normative for the HIP engine):
 *   - gr_complex = float32 pairs at every block boundary.
 *   - channel filter: causal direct form, one fmaf chain per output in tap
 *     order k = 0..ntaps-1 (bit-reproducible on the GPU).
 *   - the three moving sums of ofdm_sync_pn are accumulated in Q23.40 fixed
 *     point (error <= 2^-41 per term, far below GR's own float32 accumulate),
 *     which makes them independent of evaluation order, so a parallel GPU
 *     scan gives the same bits.  0/0 in the metric (silent input) yields 0
 *     instead of GR's NaN, and M is clamped to 1024.
 *   - the NCO phase is the closed form of gr_frequency_modulator_fc in float64
 *     (GR's float32 accumulator with its scheduler-dependent fmod wrap is not
 *     reproducible by anyone).
 *   - the mapper's rand()%arity fill is a counter-based hash of
 *     (pad_seed, packet, carrier slot).
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ofdm_hip.h"
#include "ofdm_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* small helpers                                                            */
/* ------------------------------------------------------------------------ */

static inline ofdm_c32 c32(float re, float im) {
  ofdm_c32 z = {re, im};
  return z;
}
/* gr_complex multiply as volk/SSE does it: two products per part, one add, no fma */
static inline ofdm_c32 cmul(ofdm_c32 a, ofdm_c32 b) {
  float re = a.re * b.re - a.im * b.im;
  float im = a.re * b.im + a.im * b.re;
  return c32(re, im);
}
static inline ofdm_c32 cmul_conj(ofdm_c32 a, ofdm_c32 b) { /* a * conj(b) */
  float re = a.re * b.re + a.im * b.im;
  float im = a.im * b.re - a.re * b.im;
  return c32(re, im);
}
static inline ofdm_c32 cdiv(ofdm_c32 a, ofdm_c32 b) {
  float den = b.re * b.re + b.im * b.im;
  float re = (a.re * b.re + a.im * b.im) / den;
  float im = (a.im * b.re - a.re * b.im) / den;
  return c32(re, im);
}
static inline float cnorm(ofdm_c32 a) { return a.re * a.re + a.im * a.im; }

/* complex_to_arg for the sample-and-held fine-frequency estimate.  Written out in plain float32
 * operations (Cephes-style atanf: two range reductions + a degree-9 odd polynomial, ~2 ulp) so
 * that the GPU evaluates the very same operations and gets the same bits: the NCO integrates this
 * angle over thousands of samples, which would turn a 1-ulp libm/ocml difference into 1e-4 rad. */
static inline float det_atanf_pos(float x) {
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
float orc_atan2f(float y, float x) {
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966f;
    if (y < 0.0f) return -1.5707963267948966f;
    return 0.0f;
  }
  float a = det_atanf_pos(fabsf(y / x));
  float r = (x > 0.0f) ? a : (3.14159265358979323846f - a);
  return (y < 0.0f) ? -r : r;
}

static int ilog2_ceil(unsigned v) {
  int n = 0;
  while ((1u << n) < v) n++;
  return n;
}

int orc_nbits(const ofdm_cfg *cfg) { return ilog2_ceil(cfg->arity); }

typedef struct {
  void *p;
  size_t n, cap, esz;
} vec;
static void vec_init(vec *v, size_t esz) {
  v->p = NULL;
  v->n = v->cap = 0;
  v->esz = esz;
}
static void *vec_push(vec *v, size_t count) {
  if (v->n + count > v->cap) {
    size_t nc = v->cap ? v->cap * 2 : 256;
    while (nc < v->n + count) nc *= 2;
    v->p = realloc(v->p, nc * v->esz);
    if (!v->p) abort();
    v->cap = nc;
  }
  void *r = (char *)v->p + v->n * v->esz;
  v->n += count;
  return r;
}
static void vec_free(vec *v) {
  free(v->p);
  v->p = NULL;
  v->n = v->cap = 0;
}

/* ------------------------------------------------------------------------ */
/* CRC-32  (digital_crc32 / crc.gen_and_append_crc32, ofdm_packet_utils.py:25,120,184;
 * docstring digital_swig.py:3151-3169): reflected 0xEDB88320, init/xorout ~0   */
/* ------------------------------------------------------------------------ */
static uint32_t crc_table[256];
static int crc_table_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    crc_table[i] = c;
  }
  crc_table_ready = 1;
}
uint32_t orc_crc32(const uint8_t *buf, uint64_t len) {
  if (!crc_table_ready) crc_init();
  uint32_t c = 0xFFFFFFFFu;
  for (uint64_t i = 0; i < len; i++) c = crc_table[(c ^ buf[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------ */
/* make_packet / unmake_packet  (ofdm_packet_utils.py:84-143,169-191)        */
/* ------------------------------------------------------------------------ */

/* _npadding_bytes(pkt_byte_len, samples_per_symbol=1, bits_per_symbol=1) as
 * ofdm_mod.send_pkt calls it (ofdm.py:144): byte_modulus = lcm(128/8,1)*1/1 = 16 */
static uint32_t npadding_bytes(uint32_t pkt_byte_len) {
  uint32_t r = pkt_byte_len % 16u;
  return r == 0 ? 0 : 16u - r;
}

int orc_framed_len(const ofdm_cfg *cfg, uint32_t payload_len, uint32_t *out) {
  uint32_t L = payload_len + 4; /* payload + CRC */
  if (L > OFDM_MASK_LEN) return OFDM_E_INVAL; /* ofdm_packet_utils.py:123-126 */
  uint32_t n = 4 + L + 1;       /* header + body + 0x55 */
  if (cfg->flags & OFDM_F_PAD_FOR_USRP) n += npadding_bytes(n);
  *out = n;
  return OFDM_OK;
}

int orc_make_packet(const ofdm_cfg *cfg, const uint8_t *payload, uint32_t len, uint8_t *out,
                    uint32_t *outlen) {
  uint32_t n;
  int rc = orc_framed_len(cfg, len, &n);
  if (rc) return rc;
  uint32_t off = cfg->whitener_offset;
  if (off > 15) return OFDM_E_INVAL; /* ofdm_packet_utils.py:117-118 (intent) */
  uint32_t L = len + 4;
  /* make_header: (off & 0xf) << 12 | (L & 0xfff), twice, big endian (ofdm_packet_utils.py:93-97) */
  uint32_t val = ((off & 0xF) << 12) | (L & 0x0FFF);
  out[0] = (uint8_t)(val >> 8);
  out[1] = (uint8_t)val;
  out[2] = out[0];
  out[3] = out[1];
  uint8_t *body = out + 4;
  memcpy(body, payload, len);
  uint32_t crc = orc_crc32(payload, len);
  body[len + 0] = (uint8_t)(crc >> 24); /* struct.pack(">I", crc) */
  body[len + 1] = (uint8_t)(crc >> 16);
  body[len + 2] = (uint8_t)(crc >> 8);
  body[len + 3] = (uint8_t)crc;
  for (uint32_t i = L; i < n - 4; i++) body[i] = 0x55; /* tail + USRP pad (:129,132-134) */
  /* whiten(pkt_dt, o): XOR with mask[o : o+len] (ofdm_packet_utils.py:84-87); header is not whitened.
   * numpy slicing truncates at the mask end: bytes past it would raise a shape error in the
   * reference; they cannot occur for L <= 4095 + pad within 4096+... we guard instead.      */
  for (uint32_t i = 0; i < n - 4; i++) {
    uint32_t m = off + i;
    if (m >= OFDM_MASK_LEN) return OFDM_E_INVAL;
    body[i] ^= cfg->whitening_mask[m];
  }
  *outlen = n;
  return OFDM_OK;
}

/* unmake_packet(msg, whitener_offset=0, dewhitening=1) + crc.check_crc32
 * (ofdm_packet_utils.py:169-191; called without offset at ofdm.py:303)      */
int orc_unmake_packet(const ofdm_cfg *cfg, const uint8_t *msg, uint32_t len, uint8_t *payload_out,
                      uint32_t *payload_len, int *ok) {
  if (len > OFDM_MASK_LEN) return OFDM_E_INVAL;
  if (len < 4) { /* check_crc32: len < 4 -> (False, '') */
    *ok = 0;
    *payload_len = 0;
    return OFDM_OK;
  }
  uint8_t tmp[OFDM_MASK_LEN];
  for (uint32_t i = 0; i < len; i++) tmp[i] = msg[i] ^ cfg->whitening_mask[i];
  uint32_t crc = orc_crc32(tmp, len - 4);
  uint32_t got = ((uint32_t)tmp[len - 4] << 24) | ((uint32_t)tmp[len - 3] << 16) |
                 ((uint32_t)tmp[len - 2] << 8) | (uint32_t)tmp[len - 1];
  *ok = (crc == got);
  *payload_len = len - 4;
  memcpy(payload_out, tmp, len - 4);
  return OFDM_OK;
}

/* ------------------------------------------------------------------------ */
/* subcarrier map  (digital_ofdm_mapper_bcv ctor at ofdm.py:106 with
 * container = fft_length; digital_ofdm_frame_sink ctor at ofdm.py:240 with
 * container = occupied_tones).  Default carrier string "FE7F"
 * (transmit_path.py:64 default, reset_carrier_map commented out :67); any other
 * hex string (cfg->carrier_map) goes through the same growth / centring rule.  */
/* ------------------------------------------------------------------------ */
/* sink = 0: digital_ofdm_mapper_bcv's rule (bin 4*(i+pad)+j of the fft_length bins [A.1]);
 * sink = 1: digital_ofdm_frame_sink's rule (carrier 4*i + j - diff_left of the occupied block, over the first
 *           occ/4 + diff_left digits -- its constructor's loop bound [A.10]).                                  */
int orc_carrier_map2(int occ, int container, const char *carriers, int sink, int *map, int cap) {
  /* hex digit values, MSB = lowest carrier of the nibble */
  int digits[OFDM_MAX_CARRIER_HEX + OFDM_MAX_FFT / 4 + 8];
  int nd = 0;
  if (occ < 16 || occ > OFDM_MAX_FFT || container > OFDM_MAX_FFT) return OFDM_E_INVAL;
  if (!carriers || !carriers[0]) carriers = "FE7F";
  size_t len = strlen(carriers);
  if (len > OFDM_MAX_CARRIER_HEX) return OFDM_E_INVAL;
  /* the ctor's loop: while (diff > 7) { carriers = "f" + carriers + "f"; diff -= 8; } then a
   * final partial nibble split ceil(diff/2) left, the rest right */
  int diff = occ - 4 * (int)len;
  int nf = 0;
  while (diff > 7) {
    nf++;
    diff -= 8;
  }
  int have_extra = diff > 0;
  int dl = 0, dr = 0;
  if (have_extra) {
    dl = (diff + 1) / 2; /* ceil(diff/2) */
    dr = diff - dl;
    digits[nd++] = (1 << dl) - 1;
  }
  for (int i = 0; i < nf; i++) digits[nd++] = 0xF;
  for (size_t i = 0; i < len; i++) {
    char c = carriers[i];
    int v;
    if (c >= '0' && c <= '9') v = c - '0';
    else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') v = c - 'A' + 10;
    else return OFDM_E_INVAL;
    digits[nd++] = v;
  }
  for (int i = 0; i < nf; i++) digits[nd++] = 0xF;
  if (have_extra) digits[nd++] = 0xF ^ ((1 << dr) - 1);
  int n = 0;
  if (sink) {
    /* for(i = 0; i < (d_occupied_carriers/4)+diff_left; i++) ... push_back(4*i + j - diff_left) */
    int nread = occ / 4 + dl;
    for (int i = 0; i < nread; i++) {
      int d = i < nd ? digits[i] : 0; /* past the end of the string: strtol of the terminator = 0 */
      for (int j = 0; j < 4; j++)
        if ((d >> (3 - j)) & 1) {
          int idx = 4 * i + j - dl;
          if (idx < 0 || idx >= occ) return OFDM_E_INVAL;
          if (n >= cap) return OFDM_E_CAPACITY;
          map[n++] = idx;
        }
    }
  } else {
    int pad = (container / 4 - nd) / 2; /* C integer division, as the C++ does */
    for (int i = 0; i < nd; i++)
      for (int j = 0; j < 4; j++)
        if ((digits[i] >> (3 - j)) & 1) {
          int idx = 4 * (i + pad) + j;
          if (idx < 0 || idx >= container) return OFDM_E_INVAL;
          if (n >= cap) return OFDM_E_CAPACITY;
          map[n++] = idx;
        }
  }
  if (n > occ) return OFDM_E_INVAL; /* "subcarriers allocated exceeds size of occupied carriers" */
  if (n == 0) return OFDM_E_INVAL;
  return n;
}
/* test-suite convention: container == occupied_tones asks for the frame sink's map */
int orc_carrier_map(int occ, int container, const char *carriers, int *map, int cap) {
  return orc_carrier_map2(occ, container, carriers, container == occ, map, cap);
}

/* ------------------------------------------------------------------------ */
/* FFT: gr_fft_vcc over FFTW3f (ofdm.py:112, ofdm_receiver.py~:126): unnormalised
 * DFT, float32.  FFTW's codelets and plan are not reproducible by anyone else, so
 * the NORMATIVE transform of this code base is fixed here, operation by operation,
 * and the HIP engine (csrc/fft.h) evaluates the identical expression DAG -- every
 * transform output is therefore bit-exact between the two:
 *   Stockham autosort, passes of radix 8 preceded by ONE pass of radix 2 or 4 when
 *   log2(n) is not a multiple of 3; pass with radix R over sub-transforms of length
 *   LS (LS = product of the earlier radices): butterfly j < n/R takes
 *   v[q] = src[j + q*n/R], multiplies v[q] (q >= 1) by tw[(j % LS) * q * n/(LS*R)]
 *   with ONE fused multiply-add per part (fmaf(a.re,w.re,-(a.im*w.im)),
 *   fmaf(a.re,w.im,a.im*w.re)), applies the radix-R kernel below (plain float32
 *   adds; the 1/sqrt(2) rotations are two products each) and stores
 *   dst[(j - j%LS)*R + j%LS + r*LS] = v[r].  Twiddles: (float)cos / (float)sin of
 *   -2 pi k / n evaluated in float64; the inverse transform conjugates them.      */
/* ------------------------------------------------------------------------ */
typedef struct {
  int n;
  ofdm_c32 *tw;  /* exp(-2 pi i k / n), k < n */
  ofdm_c32 *buf; /* scratch, n points */
} fft_plan;

static void fft_plan_init(fft_plan *p, int n) {
  p->n = n;
  p->tw = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)n);
  p->buf = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)n);
  for (int k = 0; k < n; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    p->tw[k] = c32((float)cos(a), (float)sin(a));
  }
}
static void fft_plan_free(fft_plan *p) {
  free(p->tw);
  free(p->buf);
}
static inline ofdm_c32 cadd(ofdm_c32 a, ofdm_c32 b) { return c32(a.re + b.re, a.im + b.im); }
static inline ofdm_c32 csub(ofdm_c32 a, ofdm_c32 b) { return c32(a.re - b.re, a.im - b.im); }
/* forward: a * (-i); inverse: a * (+i) */
static inline ofdm_c32 mul_mi(ofdm_c32 a, int inv) { return inv ? c32(-a.im, a.re) : c32(a.im, -a.re); }
static inline ofdm_c32 cmul_f(ofdm_c32 a, ofdm_c32 b) {
  return c32(fmaf(a.re, b.re, -(a.im * b.im)), fmaf(a.re, b.im, a.im * b.re));
}
static void dft8(ofdm_c32 v[8], int inv) {
  const float h = 0.70710678118654752440f;
  ofdm_c32 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  ofdm_c32 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  ofdm_c32 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  ofdm_c32 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  if (inv) {
    a5 = c32((a5.re - a5.im) * h, (a5.re + a5.im) * h);
    a7 = c32((-a7.re - a7.im) * h, (a7.re - a7.im) * h);
  } else {
    a5 = c32((a5.re + a5.im) * h, (a5.im - a5.re) * h);
    a7 = c32((a7.im - a7.re) * h, (-a7.re - a7.im) * h);
  }
  a6 = mul_mi(a6, inv);
  ofdm_c32 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  ofdm_c32 b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3), inv);
  ofdm_c32 c0 = cadd(a4, a6), c2 = csub(a4, a6);
  ofdm_c32 c1 = cadd(a5, a7), c3 = mul_mi(csub(a5, a7), inv);
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(c0, c1);
  v[5] = csub(c0, c1);
  v[3] = cadd(c2, c3);
  v[7] = csub(c2, c3);
}
static void dft4(ofdm_c32 v[4], int inv) {
  ofdm_c32 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
  ofdm_c32 s1 = cadd(v[1], v[3]), d1 = mul_mi(csub(v[1], v[3]), inv);
  v[0] = cadd(s0, s1);
  v[2] = csub(s0, s1);
  v[1] = cadd(d0, d1);
  v[3] = csub(d0, d1);
}
static void fft_pass(const fft_plan *p, const ofdm_c32 *src, ofdm_c32 *dst, int R, int LS, int inv) {
  int n = p->n, stride = n / R, tws = n / (LS * R);
  for (int j = 0; j < stride; j++) {
    ofdm_c32 v[8];
    int k = j % LS;
    for (int q = 0; q < R; q++) v[q] = src[j + q * stride];
    if (LS > 1)
      for (int q = 1; q < R; q++) {
        ofdm_c32 w = p->tw[k * q * tws];
        if (inv) w.im = -w.im;
        v[q] = cmul_f(v[q], w);
      }
    if (R == 8) {
      dft8(v, inv);
    } else if (R == 4) {
      dft4(v, inv);
    } else {
      ofdm_c32 sum = cadd(v[0], v[1]), dif = csub(v[0], v[1]);
      v[0] = sum;
      v[1] = dif;
    }
    int obase = (j - k) * R + k;
    for (int r = 0; r < R; r++) dst[obase + r * LS] = v[r];
  }
}
/* in-place; inverse != 0 uses exp(+...); no scaling either way.  n = 2^k >= 8 */
static void fft_exec(const fft_plan *p, ofdm_c32 *x, int inverse) {
  int n = p->n, logn = ilog2_ceil((unsigned)n);
  ofdm_c32 *a = x, *b = p->buf;
  int LS = 1;
  int lead = (logn % 3 == 1) ? 2 : (logn % 3 == 2) ? 4 : 0;
  if (lead) {
    fft_pass(p, a, b, lead, LS, inverse);
    LS *= lead;
    ofdm_c32 *t = a; a = b; b = t;
  }
  while (LS < n) {
    fft_pass(p, a, b, 8, LS, inverse);
    LS *= 8;
    ofdm_c32 *t = a; a = b; b = t;
  }
  if (a != x) memcpy(x, a, sizeof(ofdm_c32) * (size_t)n);
}

/* ------------------------------------------------------------------------ */
/* Deterministic elementary functions.  libm (CPU) and ocml (GPU) differ in the
 * last bit, which is enough to make two float32 receivers disagree on a slicer
 * boundary.  The engine evaluates the same plain IEEE operations (fma() where
 * written, every other operation separately rounded), so these have the same
 * bits on both sides.  Accuracy is that of a good libm (tests check <= 2 ulp). */
/* ------------------------------------------------------------------------ */
/* sin and cos of a float32 argument (|x| < ~1e6): reduction by multiples of pi/2 in
 * float64 (two-part pi/2), Cephes sinf/cosf polynomials on [-pi/4, pi/4].      */
void orc_sincosf(float x, float *sn, float *cs) {
  double xd = (double)x;
  double kd = rint(xd * 0.63661977236758134308);
  double yd = fma(-kd, 1.57079632673412561417e+00, xd);
  yd = fma(-kd, 6.07710050650619224932e-11, yd);
  float y = (float)yd;
  int q = (int)kd & 3;
  float z = y * y;
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
  float s_, c_;
  if (q & 1) {
    s_ = pc;
    c_ = -ps;
  } else {
    s_ = ps;
    c_ = pc;
  }
  if (q & 2) {
    s_ = -s_;
    c_ = -c_;
  }
  *sn = s_;
  *cs = c_;
}
/* exp(j ph) in float64 for the NCO: ph is first wrapped to [-pi, pi] exactly as
 * written, then reduced by pi/2 and evaluated with the fdlibm kernel polynomials. */
typedef struct {
  double re, im;
} dcx;
dcx orc_expj(double ph) {
  ph = ph - 6.283185307179586476925 * floor(ph / 6.283185307179586476925 + 0.5);
  double kd = rint(ph * 0.63661977236758134308);
  double y = fma(-kd, 1.57079632673412561417e+00, ph);
  y = fma(-kd, 6.07710050650619224932e-11, y);
  int q = (int)kd & 3;
  double z = y * y;
  double r = 1.58969099521155010221e-10;
  r = fma(z, r, -2.50507602534068634195e-08);
  r = fma(z, r, 2.75573137070700676789e-06);
  r = fma(z, r, -1.98412698298579493134e-04);
  r = fma(z, r, 8.33333333332248946124e-03);
  r = fma(z, r, -1.66666666666666324348e-01);
  double sn = fma(y * z, r, y);
  double c = -1.13596475577881948265e-11;
  c = fma(z, c, 2.08757232129817482790e-09);
  c = fma(z, c, -2.75573143513906633035e-07);
  c = fma(z, c, 2.48015872894767294178e-05);
  c = fma(z, c, -1.38888888888741095749e-03);
  c = fma(z, c, 4.16666666666666019037e-02);
  double cs = fma(z * z, c, fma(-0.5, z, 1.0));
  dcx o;
  if (q & 1) {
    o.im = cs;
    o.re = -sn;
  } else {
    o.im = sn;
    o.re = cs;
  }
  if (q & 2) {
    o.im = -o.im;
    o.re = -o.re;
  }
  return o;
}
static inline dcx dmul(dcx a, dcx b) {
  dcx r;
  r.re = a.re * b.re - a.im * b.im;
  r.im = a.re * b.im + a.im * b.re;
  return r;
}

/* array forms for the test-suite (tests/test_oracle.py checks them against libm / numpy.fft) */
void orc_sincosf_vec(const float *x, uint64_t n, float *sn, float *cs) {
  for (uint64_t i = 0; i < n; i++) orc_sincosf(x[i], &sn[i], &cs[i]);
}
void orc_expj_vec(const double *ph, uint64_t n, double *re, double *im) {
  for (uint64_t i = 0; i < n; i++) {
    dcx z = orc_expj(ph[i]);
    re[i] = z.re;
    im[i] = z.im;
  }
}
int orc_fft(ofdm_c32 *x, int n, int inverse) {
  if (n < 8 || (n & (n - 1))) return OFDM_E_INVAL;
  fft_plan p;
  fft_plan_init(&p, n);
  fft_exec(&p, x, inverse);
  fft_plan_free(&p);
  return OFDM_OK;
}

/* Sum of per-lane partial sums the way the engine's workgroup of T = N/8 threads adds them: groups of 64
 * lanes by a butterfly (distance 32, 16, .. 1; float addition is commutative, so every lane ends with the
 * same value), the groups then in order; fewer than 64 lanes: in order.  Normative for the two float32
 * reductions of the receiver (coarse-offset correlation, PLL error) -- a pairwise tree, at least as accurate
 * as GNU Radio's sequential loop.                                                                      */
static float lane_tree_sum(const float *part, int T) {
  if (T < 64) {
    float s = 0.0f;
    for (int i = 0; i < T; i++) s = s + part[i];
    return s;
  }
  float s = 0.0f;
  for (int w = 0; w < T / 64; w++) {
    float v[64], u[64];
    for (int l = 0; l < 64; l++) v[l] = part[w * 64 + l];
    for (int d = 32; d > 0; d >>= 1) {
      for (int l = 0; l < 64; l++) u[l] = v[l] + v[l ^ d];
      for (int l = 0; l < 64; l++) v[l] = u[l];
    }
    if (T == 64) return v[0];
    s = s + v[0];
  }
  return s;
}

/* ------------------------------------------------------------------------ */
/* TX                                                                       */
/* ------------------------------------------------------------------------ */

/* symbols the mapper emits for one framed packet, excluding the preamble:
 * a new symbol is started while message bytes remain (digital_ofdm_mapper_bcv::work) */
uint32_t orc_tx_data_symbols(const ofdm_cfg *cfg, uint32_t framed_len, int ncarriers) {
  uint64_t bits = 8ull * framed_len;
  uint64_t per = (uint64_t)ncarriers * (uint64_t)orc_nbits(cfg);
  return (uint32_t)((bits + per - 1) / per);
}

/* counter-based replacement for the mapper's  rand() % arity  fill */
uint32_t orc_pad_symbol(uint64_t seed, uint64_t pkt, uint64_t slot, uint32_t arity) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (pkt + 1) + 0xBF58476D1CE4E5B9ull * (slot + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)((z >> 32) % arity);
}

/* chunk c (nbits wide) of the little-endian-within-byte bit stream of msg, or -1
 * when the message cannot supply all nbits (incomplete chunk is dropped)       */
static int msg_chunk(const uint8_t *msg, uint32_t len, uint64_t c, int nbits) {
  uint64_t b0 = c * (uint64_t)nbits;
  if (b0 + (uint64_t)nbits > 8ull * len) return -1;
  unsigned v = 0;
  for (int k = 0; k < nbits; k++) {
    uint64_t b = b0 + (uint64_t)k;
    v |= (unsigned)((msg[b >> 3] >> (b & 7)) & 1) << k;
  }
  return (int)v;
}

int orc_tx(const ofdm_cfg *cfg, const uint8_t *payloads, const uint64_t *payload_off,
           const uint32_t *payload_len, int npkt, uint64_t lead, ofdm_c32 *iq_out, uint64_t iq_cap,
           uint64_t *nsamples_out, ofdm_c32 *freq_tap, uint8_t *framed_tap, uint64_t *framed_off_tap) {
  return orc_tx_ex(cfg, payloads, payload_off, payload_len, npkt, lead, iq_out, iq_cap, nsamples_out, freq_tap, framed_tap,
                   framed_off_tap, NULL);
}
/* ifft_tap: [nsym][N], the transform's output vectors before the cyclic prefix (ofdm_ifft_c.dat, ofdm.py:128) */
int orc_tx_ex(const ofdm_cfg *cfg, const uint8_t *payloads, const uint64_t *payload_off,
              const uint32_t *payload_len, int npkt, uint64_t lead, ofdm_c32 *iq_out, uint64_t iq_cap,
              uint64_t *nsamples_out, ofdm_c32 *freq_tap, uint8_t *framed_tap, uint64_t *framed_off_tap, ofdm_c32 *ifft_tap) {
  int N = (int)cfg->fft_length, CP = (int)cfg->cp_length, occ = (int)cfg->occupied_tones;
  int L = N + CP;
  int nbits = orc_nbits(cfg);
  if (occ > N) return OFDM_E_INVAL; /* mapper ctor: occupied_carriers > fft_length */
  int *map = (int *)malloc(sizeof(int) * (size_t)occ);
  int nc = orc_carrier_map2(occ, N, cfg->carrier_map, 0, map, occ);
  if (nc < 0) {
    free(map);
    return nc;
  }
  int zl = (N - occ + 1) / 2; /* ceil((N-occ)/2), ofdm.py:71 */
  /* padded preamble (ofdm.py:83-87) */
  ofdm_c32 *pre = (ofdm_c32 *)calloc((size_t)N, sizeof(ofdm_c32));
  for (int i = 0; i < occ; i++) pre[zl + i] = cfg->known_symbol[i];

  fft_plan plan;
  fft_plan_init(&plan, N);
  ofdm_c32 *sym = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)N);
  ofdm_c32 *tmp = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)N);
  uint8_t *pkt = (uint8_t *)malloc(OFDM_MASK_LEN + 64);
  float scale1 = (float)(1.0 / sqrt((double)N)); /* gr.multiply_const_cc(1.0/math.sqrt(N)), ofdm.py:114 */
  float amp = cfg->tx_amplitude;                 /* transmit_path.amp, transmit_path.py:48-54 */

  uint64_t pos = lead, nsym_total = 0, foff = 0;
  int rc = OFDM_OK;
  for (int p = 0; p < npkt && rc == OFDM_OK; p++) {
    uint32_t plen;
    rc = orc_make_packet(cfg, payloads + payload_off[p], payload_len[p], pkt, &plen);
    if (rc) break;
    if (framed_tap) {
      memcpy(framed_tap + foff, pkt, plen);
      framed_off_tap[p] = foff;
    }
    foff += plen;
    uint32_t nds = orc_tx_data_symbols(cfg, plen, nc);
    for (uint32_t s = 0; s <= nds; s++) {
      /* s == 0: ofdm_insert_preamble emits the preamble ahead of the flagged symbol (ofdm.py:111) */
      if (s == 0) {
        memcpy(sym, pre, sizeof(ofdm_c32) * (size_t)N);
      } else {
        memset(sym, 0, sizeof(ofdm_c32) * (size_t)N);
        for (int i = 0; i < nc; i++) {
          uint64_t slot = (uint64_t)(s - 1) * (uint64_t)nc + (uint64_t)i;
          int bits = msg_chunk(pkt, plen, slot, nbits);
          if (bits < 0) bits = (int)orc_pad_symbol(cfg->pad_seed, (uint64_t)p, slot, cfg->arity);
          sym[map[i]] = cfg->constellation[bits];
        }
      }
      if (freq_tap) memcpy(freq_tap + nsym_total * (uint64_t)N, sym, sizeof(ofdm_c32) * (size_t)N);
      /* gr.fft_vcc(N, False, [], True): swap input halves, unnormalised inverse DFT (ofdm.py:112) */
      for (int k = 0; k < N; k++) tmp[k] = sym[(k + N / 2) % N];
      fft_exec(&plan, tmp, 1);
      if (ifft_tap) memcpy(ifft_tap + nsym_total * (uint64_t)N, tmp, sizeof(ofdm_c32) * (size_t)N);
      if (pos + (uint64_t)L > iq_cap) {
        rc = OFDM_E_CAPACITY;
        break;
      }
      /* ofdm_cyclic_prefixer(N, N+CP) (ofdm.py:113) then the two multiply_const_cc */
      ofdm_c32 *o = iq_out + pos;
      for (int n = 0; n < L; n++) {
        ofdm_c32 v = tmp[(n + N - CP) % N];
        v.re = v.re * scale1;
        v.im = v.im * scale1;
        v.re = v.re * amp;
        v.im = v.im * amp;
        o[n] = v;
      }
      pos += (uint64_t)L;
      nsym_total++;
    }
  }
  if (framed_tap) framed_off_tap[npkt] = foff;
  *nsamples_out = pos;
  free(pkt);
  free(tmp);
  free(sym);
  fft_plan_free(&plan);
  free(pre);
  free(map);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* synthetic channel (stands in for the UHD sink/source pair)                */
/* ------------------------------------------------------------------------ */
/* Philox-2x32-7 (Salmon et al., SC'11: Random123's reduced-round variant), key = stream key.  The channel draws one call per PAIR of samples
 * (counter = sample index / 2): word 0 serves the even sample, word 1 the odd one; a sample's word gives 16 bits
 * of Box-Muller radius and 16 bits of angle. */
static inline uint32_t chan_key(uint64_t seed, uint64_t stream) {
  return (uint32_t)seed ^ (uint32_t)(seed >> 32) ^ ((uint32_t)stream * 0x9E3779B9u + (uint32_t)(stream >> 32) * 0x85EBCA6Bu);
}
#define CHAN_PHILOX_ROUNDS 7
void orc_philox_r(uint64_t seed, uint64_t stream, uint64_t idx, int rounds, uint32_t out[2]) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), k = chan_key(seed, stream);
  for (int r = 0; r < rounds; r++) {
    uint64_t p = (uint64_t)0xD256D193u * c0;
    uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
    c0 = hi ^ k ^ c1;
    c1 = lo;
    k += 0x9E3779B9u;
  }
  out[0] = c0;
  out[1] = c1;
}
void orc_philox(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t out[2]) {
  orc_philox_r(seed, stream, idx, CHAN_PHILOX_ROUNDS, out);
}

int orc_channel(ofdm_c32 *iq, uint64_t n, const ofdm_chan *ch, uint64_t index0) {
  const float inv16 = 1.0f / 65536.0f;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t idx = index0 + i;
    ofdm_c32 x = iq[i];
    if (ch->cfo != 0.0f) {
      double ph = (double)ch->cfo * (double)idx;
      ph = ph - 2.0 * M_PI * floor(ph / (2.0 * M_PI) + 0.5);
      ofdm_c32 r = c32((float)cos(ph), (float)sin(ph));
      x = cmul(x, r);
    }
    if (ch->sigma > 0.0f) {
      uint32_t r[2];
      orc_philox(ch->seed, ch->stream_id, idx >> 1, r);
      uint32_t word = r[idx & 1];
      float u1 = ((float)(word >> 16) + 0.5f) * inv16;
      float u2 = ((float)(word & 0xFFFFu) + 0.5f) * inv16;
      float rad = sqrtf(-2.0f * logf(u1));
      float th = 6.28318530717958647692f * u2;
      float s = ch->sigma * 0.70710678118654752440f;
      x.re = x.re + s * (rad * cosf(th));
      x.im = x.im + s * (rad * sinf(th));
    }
    iq[i] = x;
  }
  return OFDM_OK;
}

/* ------------------------------------------------------------------------ */
/* RX                                                                       */
/* ------------------------------------------------------------------------ */
struct orc_rx_result {
  ofdm_stats st;
  uint32_t tap_mask;
  int N, occ;
  vec y;        /* c32 */
  vec metric;   /* f32 */
  vec peaks;    /* u64 */
  vec peaks_gr; /* u64: flags of the literal float32 recurrence (cross-check, ORC_TAP_PEAKS_GR) */
  vec presel;   /* f32 [nsamples]: float32 pre-selection metric (OFDM_TAP_RX_PRESEL) */
  vec ranges;   /* i32 [ntiles][2]: exact-evaluation range of every 2048-sample tile, -1 -1 = none (ORC_TAP_RANGES) */
  uint64_t presel_miss; /* samples with exact u > theta outside every range (must be 0: see orc_rx_presel_miss) */
  vec angles;   /* f32 */
  vec frames;   /* u64 x2 */
  vec fft;      /* c32 */
  vec acq;      /* c32 */
  vec sink;     /* c32 */
  vec sampler;  /* c32: the sampled, derotated symbols (FFT input) */
  vec sigmix;   /* c32 [nsamples] */
  vec nco;      /* c32 [nsamples] */
  vec raw;      /* u8: frame sink messages, concatenated */
  vec raw_off;  /* u64 */
  vec payload;  /* u8 */
  vec pay_off;  /* u64 (n+1) */
  vec pay_len;  /* u32 */
  vec pay_ok;   /* u8 */
};

#define QSCALE 1099511627776.0 /* 2^40 */
#define QINV (1.0 / 1099511627776.0)

static inline int64_t q40(float v) {
  /* |v| is clamped so that 2048-term windows stay inside int64 */
  if (v > 511.0f) v = 511.0f;
  if (v < -511.0f) v = -511.0f;
  return (int64_t)llrint((double)v * QSCALE);
}

/* gr_fft_filter_ccc(1, taps) (ofdm_receiver.py~:76,131), the way GNU Radio runs it [SURVEY A.5]: overlap-save
 * with transform length F = 2 * 2^ceil(log2(ntaps)) (at least 64 here) and B = F - ntaps + 1 outputs per block on
 * the grid b*B of the stream.  Block b transforms the F samples x[b*B - (ntaps-1) .. b*B + B) (zero before the
 * stream starts), multiplies bin by bin with the transformed taps -- taps zero-padded to F, transformed, scaled by
 * 1/F; computed in float64 and rounded once, where GR rounds FFTW's float32 result -- as volk does (two products
 * and one add per part), transforms back (unnormalised) and keeps the last B points.  Result == causal linear
 * convolution with zero pre-history, to float32 rounding.  A final partial block is processed with zeros after
 * the end of the input (GR would wait for more samples).  The transform is fft_exec above.                     */
int orc_filter_fft_len(int ntaps) {
  int p2 = 1;
  while (p2 < ntaps) p2 <<= 1;
  int f = 2 * p2;
  return f < 64 ? 64 : f;
}
static void chan_filter(const ofdm_cfg *cfg, const ofdm_c32 *x, uint64_t n, ofdm_c32 *y) {
  int nt = (int)cfg->ntaps;
  int F = orc_filter_fft_len(nt), B = F - nt + 1, ntm1 = nt - 1;
  fft_plan plan;
  fft_plan_init(&plan, F);
  ofdm_c32 *H = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)F);
  ofdm_c32 *w = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)F);
  {
    double *cs = (double *)malloc(sizeof(double) * (size_t)F), *sn = (double *)malloc(sizeof(double) * (size_t)F);
    for (int m = 0; m < F; m++) {
      double a = -2.0 * M_PI * (double)m / (double)F;
      cs[m] = cos(a);
      sn[m] = sin(a);
    }
    for (int k = 0; k < F; k++) {
      double re = 0.0, im = 0.0;
      for (int i = 0; i < nt; i++) {
        int idx = (int)(((long long)k * i) % F);
        re = re + (double)cfg->taps[i] * cs[idx];
        im = im + (double)cfg->taps[i] * sn[idx];
      }
      H[k] = c32((float)(re / (double)F), (float)(im / (double)F));
    }
    free(cs);
    free(sn);
  }
  for (uint64_t b0 = 0; b0 < n; b0 += (uint64_t)B) {
    for (int i = 0; i < F; i++) {
      int64_t xi = (int64_t)b0 - ntm1 + i;
      w[i] = (xi >= 0 && (uint64_t)xi < n) ? x[xi] : c32(0.0f, 0.0f);
    }
    fft_exec(&plan, w, 0);
    for (int k = 0; k < F; k++) w[k] = cmul(w[k], H[k]);
    fft_exec(&plan, w, 1);
    for (int o = 0; o < B && b0 + (uint64_t)o < n; o++) y[b0 + (uint64_t)o] = w[ntm1 + o];
  }
  free(w);
  free(H);
  fft_plan_free(&plan);
}

/* ofdm_sync_pn(N, CP) (ofdm_receiver.py~:97-101): delay N/2, conj, multiply, two
 * N/2 moving sums, |.|^2, square, divide, CP-length moving average, -1.
 * Outputs u[n] (peak detector input) and P[n] (for complex_to_arg).            */
static void sync_metric(const ofdm_cfg *cfg, const ofdm_c32 *y, uint64_t n, float *u, ofdm_c32 *P) {
  int D = (int)cfg->fft_length / 2, CP = (int)cfg->cp_length;
  float tapcp = (float)(1.0 / (double)CP);
  int64_t *qpr = (int64_t *)malloc(sizeof(int64_t) * (size_t)D);
  int64_t *qpi = (int64_t *)malloc(sizeof(int64_t) * (size_t)D);
  int64_t *qr = (int64_t *)malloc(sizeof(int64_t) * (size_t)D);
  int64_t *qm = (int64_t *)malloc(sizeof(int64_t) * (size_t)CP);
  memset(qpr, 0, sizeof(int64_t) * (size_t)D);
  memset(qpi, 0, sizeof(int64_t) * (size_t)D);
  memset(qr, 0, sizeof(int64_t) * (size_t)D);
  memset(qm, 0, sizeof(int64_t) * (size_t)CP);
  int64_t spr = 0, spi = 0, sr = 0, sm = 0;
  for (uint64_t i = 0; i < n; i++) {
    ofdm_c32 a = y[i];
    ofdm_c32 d = (i >= (uint64_t)D) ? y[i - (uint64_t)D] : c32(0.0f, 0.0f);
    ofdm_c32 c = cmul_conj(a, d);
    float e = a.re * a.re + a.im * a.im;
    int s = (int)(i % (uint64_t)D);
    int64_t v;
    v = q40(c.re);
    spr += v - qpr[s];
    qpr[s] = v;
    v = q40(c.im);
    spi += v - qpi[s];
    qpi[s] = v;
    v = q40(e);
    sr += v - qr[s];
    qr[s] = v;
    float pre = (float)((double)spr * QINV);
    float pim = (float)((double)spi * QINV);
    float r = (float)((double)sr * QINV);
    float num = pre * pre + pim * pim;
    float den = r * r;
    float m = (den > 0.0f) ? (num / den) : 0.0f;
    if (!(m <= 1024.0f)) m = 1024.0f;
    int sc = (int)(i % (uint64_t)CP);
    v = (int64_t)llrint((double)m * QSCALE);
    sm += v - qm[sc];
    qm[sc] = v;
    float mbar = (float)((double)sm * QINV * (double)tapcp);
    u[i] = mbar + (-1.0f);
    P[i] = c32(pre, pim);
  }
  free(qpr);
  free(qpi);
  free(qr);
  free(qm);
}

/* gr_peak_detector_fb(0.20, 0.20, 30, 0.001) over the whole stream as one buffer, LITERALLY: the running average
 * as the float32 recurrence avg = alpha*u + (1-alpha)*avg from the first sample on.  Kept as the cross-check of the
 * normative evaluation below (ORC_TAP_PEAKS_GR): the two give the same flags unless a comparison u > avg*factor
 * falls inside the recurrence's own float32 rounding noise (~1e-6). */
static void peak_detect_gr(const ofdm_cfg *cfg, const float *u, uint64_t n, vec *peaks) {
  float rise = cfg->peak_rise, fall = cfg->peak_fall, alpha = cfg->peak_alpha;
  float one_m_alpha = 1.0f - alpha;
  float avg = 0.0f, peak_val = -INFINITY;
  uint64_t peak_ind = 0;
  int state = 0;
  uint64_t i = 0;
  while (i < n) {
    if (state == 0) {
      if (u[i] > avg * rise) {
        state = 1;
      } else {
        avg = alpha * u[i] + one_m_alpha * avg;
        i++;
      }
    } else {
      if (u[i] > peak_val) {
        peak_val = u[i];
        peak_ind = i;
        avg = alpha * u[i] + one_m_alpha * avg;
        i++;
      } else if (u[i] > avg * fall) {
        avg = alpha * u[i] + one_m_alpha * avg;
        i++;
      } else {
        *(uint64_t *)vec_push(peaks, 1) = peak_ind;
        state = 0;
        peak_val = -INFINITY;
      }
    }
  }
  /* a run still open at the end of the stream raises no flag (GR would wait for more input) */
}

/* ------------------------------------------------------------------------ */
/* gr_peak_detector_fb, NORMATIVE evaluation (what the engine runs).
 *
 * The detector's average obeys avg = alpha*u + (1-alpha)*avg on every sample whatever its state, and the state
 * machine can only act where u > theta = -max(rise, fall) (avg >= -1, so both thresholds are >= theta): maximal
 * runs {u > theta} are independent once avg at their first sample is known.  A float32 recurrence from the first
 * sample of the stream has no value a parallel machine can reproduce, so -- as for the moving sums and the
 * transforms -- ONE evaluation of that starting value is fixed here and the engine performs the same one:
 *
 *  1. Tiles of 2048 samples on the grid of the call's first sample.  A cheap float32 evaluation of the metric
 *     (u32, "pre-selection") of every tile in a fixed schedule: 256 lanes of 8 consecutive samples, the window
 *     sums P, R anchored afresh at the tile start (lane-strided partial sums over the D samples before it), lane
 *     chains + the 64-lane scan network below + the four groups in order; M = |P|^2 / max(R^2, 1e-37) (IEEE
 *     division), min(M, 1024); the CP-length mean of M the same way from the CP values before the tile.
 *  2. The samples with u32 > theta - 1e-3 of a tile span its RANGE [amin, bmax]; there u is evaluated in the
 *     normative Q23.40 arithmetic (sync_metric above), and only there can a candidate lie.
 *  3. The tile's contribution to the average, B = sum_k alpha * decay^(Tl-1-k) * u[k]: outside the range from
 *     u32, in float32 (a lane's chain fmaf(., 1-alpha, alpha*u) over its 8 samples, times decay^(samples after the
 *     lane), scan network, groups in order); inside it from the exact u as a sum of Q40-rounded float64 products
 *     (integer sum: no order).  The average before a tile: sum_{k>=1} decay^(2048 k) * B[g-k], k ascending, until
 *     the weight drops below 1e-30.
 *  4. At the first sample s of a run: avg = float32(avg_before_tile * decay^s + (Bpre + X(s)) / decay^(Tl - s)),
 *     X(s) the exact part up to s-1; from there the float32 recurrence and gr_peak_detector_fb's state machine,
 *     through following tiles while the run goes on.
 * decay^j is the table d[0] = 1, d[j] = d[j-1] * (double)(1.0f - alpha).                                        */
/* ------------------------------------------------------------------------ */
#define SY_T 2048
#define SY_NTH 256
#define SY_V 8
#define SY_GUARD 1.0e-3f
#define SY_ILL 0.015625f /* 1/64 */

typedef struct {
  uint64_t start, end; /* absolute sample indices */
  double bloc;         /* zero-initialised average over the tile's samples before start */
} sy_piece;

/* the 64-lane inclusive scan network (four shifts inside rows of 16 lanes, then two row broadcasts); every step
 * reads all its inputs before writing */
static void sy_scan64(float *v) {
  float o[64];
  static const int sh[4] = {1, 2, 4, 8};
  for (int s = 0; s < 4; s++) {
    memcpy(o, v, sizeof(o));
    for (int i = 0; i < 64; i++)
      if ((i & 15) >= sh[s]) v[i] = o[i - sh[s]] + o[i];
  }
  memcpy(o, v, sizeof(o));
  for (int i = 16; i < 32; i++) v[i] = o[15] + o[i];
  for (int i = 48; i < 64; i++) v[i] = o[47] + o[i];
  memcpy(o, v, sizeof(o));
  for (int i = 32; i < 64; i++) v[i] = o[31] + o[i];
}
/* exclusive scan of v over the 256 lanes and the total of s: the network per group of 64, the groups' totals in order */
static void sy_block_scan(const float *v, const float *s, float *excl, float *sum) {
  float inc[SY_NTH], rs[SY_NTH];
  memcpy(inc, v, sizeof(inc));
  memcpy(rs, s, sizeof(rs));
  for (int w = 0; w < SY_NTH / 64; w++) {
    sy_scan64(inc + 64 * w);
    sy_scan64(rs + 64 * w);
  }
  float sm = 0.0f;
  for (int w = 0; w < SY_NTH / 64; w++) sm += rs[64 * w + 63];
  for (int w = 0; w < SY_NTH / 64; w++) {
    float base = 0.0f;
    for (int i = 0; i < w; i++) base += inc[64 * i + 63];
    for (int l = 0; l < 64; l++) {
      int t = 64 * w + l;
      excl[t] = base + inc[t] - v[t];
    }
  }
  *sum = sm;
}
/* a * conj(w) with the placement of fused operations the engine's packed instructions have */
static inline ofdm_c32 sy_cmulc(ofdm_c32 a, ofdm_c32 w) {
  return c32(fmaf(a.re, w.re, a.im * w.im), fmaf(a.re, -w.im, a.im * w.re));
}
static inline int64_t sy_q40d(double x) { return (int64_t)llrint(x * QSCALE); }

typedef struct {
  int amin, bmax;     /* range, tile-relative; bmax < 0: none */
  float gpre, gpost;  /* float32 part of B before / after the range (no range: gpre = the whole tile) */
} sy_tile;

static void peak_detect(const ofdm_cfg *cfg, const ofdm_c32 *y, const float *ux, uint64_t n, vec *peaks, float *u32_tap,
                        int32_t *range_tap, uint64_t *presel_miss) {
  const int T = SY_T, D = (int)cfg->fft_length / 2, CP = (int)cfg->cp_length;
  const float rise = cfg->peak_rise, fall = cfg->peak_fall, alpha = cfg->peak_alpha;
  const float one_m_alpha = 1.0f - alpha, decay_f = one_m_alpha;
  const double decay = (double)one_m_alpha, alpha_d = (double)alpha;
  const float theta = -fmaxf(rise, fall), athr = theta - SY_GUARD;
  const float inv_cp = 1.0f / (float)CP;
  const uint64_t ntiles = (n + (uint64_t)T - 1) / (uint64_t)T;
  double *dpow = (double *)malloc(sizeof(double) * (size_t)(T + 1));
  dpow[0] = 1.0;
  for (int j = 1; j <= T; j++) dpow[j] = dpow[j - 1] * decay;
  float wf[SY_NTH];
  for (int t = 0; t < SY_NTH; t++) wf[t] = (float)dpow[T - SY_V * (t + 1)];
  sy_tile *tl = (sy_tile *)calloc(ntiles ? ntiles : 1, sizeof(sy_tile));
  double *B = (double *)calloc(ntiles ? ntiles : 1, sizeof(double));
  float *mh = (float *)calloc((size_t)CP, sizeof(float)); /* M of the CP samples before the tile */
  float *mt = (float *)malloc(sizeof(float) * (size_t)T);
  float *useq = (float *)malloc(sizeof(float) * (size_t)T);
#define SY_Y(ix) (((ix) >= 0 && (uint64_t)(ix) < n) ? y[(ix)] : c32(0.0f, 0.0f))
  /* ---- 1-3: pre-selection metric, range and float32 summary of every tile ---- */
  int prev_ill = 0;
  for (uint64_t g = 0; g < ntiles; g++) {
    const int64_t t0 = (int64_t)(g * (uint64_t)T);
    const int Tl = (t0 + T <= (int64_t)n) ? T : (int)((int64_t)n - t0);
    float pr[SY_NTH][SY_V], pi[SY_NTH][SY_V], pe[SY_NTH][SY_V];
    float va[SY_NTH], vb[SY_NTH], vc[SY_NTH], sa[SY_NTH], sb[SY_NTH], sc[SY_NTH];
    for (int t = 0; t < SY_NTH; t++) {
      float tr = 0.0f, ti = 0.0f, te = 0.0f;
      for (int j = 0; j < SY_V; j++) {
        const int64_t i = t0 + SY_V * t + j;
        const ofdm_c32 a = SY_Y(i), d1 = SY_Y(i - D), d2 = SY_Y(i - 2 * D);
        const ofdm_c32 pa = sy_cmulc(a, d1), pb = sy_cmulc(d1, d2);
        tr = tr + (pa.re - pb.re);
        ti = ti + (pa.im - pb.im);
        te += fmaf(a.re, a.re, a.im * a.im) - fmaf(d1.re, d1.re, d1.im * d1.im);
        pr[t][j] = tr;
        pi[t][j] = ti;
        pe[t][j] = te;
      }
      va[t] = tr;
      vb[t] = ti;
      vc[t] = te;
      /* anchor: the window sums at the sample before the tile, lane t takes the terms t, t+256, ... of the window */
      float ar = 0.0f, ai = 0.0f, ae = 0.0f;
      for (int m = -D + t; m < 0; m += SY_NTH) {
        const ofdm_c32 a = SY_Y(t0 + m), d1 = SY_Y(t0 + m - D);
        const ofdm_c32 pa = sy_cmulc(a, d1);
        ar = ar + pa.re;
        ai = ai + pa.im;
        ae += fmaf(a.re, a.re, a.im * a.im);
      }
      sa[t] = ar;
      sb[t] = ai;
      sc[t] = ae;
    }
    float ea[SY_NTH], eb[SY_NTH], ec[SY_NTH], anca, ancb, ancc;
    sy_block_scan(va, sa, ea, &anca);
    sy_block_scan(vb, sb, eb, &ancb);
    sy_block_scan(vc, sc, ec, &ancc);
    float rmn[SY_NTH], rmx[SY_NTH];
    for (int t = 0; t < SY_NTH; t++) {
      const float bx = anca + ea[t], by = ancb + eb[t], be = ancc + ec[t];
      float lo = INFINITY, hi = -INFINITY;
      for (int j = 0; j < SY_V; j++) {
        const float px = bx + pr[t][j], py = by + pi[t][j], r = be + pe[t][j];
        const float num = fmaf(px, px, py * py);
        float m = num / fmaxf(r * r, 1e-37f);
        m = fminf(m, 1024.0f);
        mt[SY_V * t + j] = m;
        lo = fminf(lo, r);
        hi = fmaxf(hi, r);
      }
      rmn[t] = lo;
      rmx[t] = hi;
    }
    /* CP-length mean of M: position i of [history | tile] holds M[n - CP] of the tile's sample i */
    float pm[SY_NTH][SY_V], vm[SY_NTH], sm_[SY_NTH], em[SY_NTH], mach;
    for (int t = 0; t < SY_NTH; t++) {
      float ms = 0.0f;
      for (int j = 0; j < SY_V; j++) {
        const int i = SY_V * t + j;
        ms += mt[i] - ((i < CP) ? mh[i] : mt[i - CP]);
        pm[t][j] = ms;
      }
      vm[t] = ms;
      float ma = 0.0f;
      for (int m = -CP + t; m < 0; m += SY_NTH) ma += mh[CP + m];
      sm_[t] = ma;
    }
    sy_block_scan(vm, sm_, em, &mach);
    for (int t = 0; t < SY_NTH; t++)
      for (int j = 0; j < SY_V; j++) useq[SY_V * t + j] = (mach + em[t] + pm[t][j]) * inv_cp - 1.0f;
    for (int i = 0; i < CP; i++) mh[i] = mt[T - CP + i];
    if (u32_tap)
      for (int i = 0; i < Tl; i++) u32_tap[t0 + i] = useq[i];
    /* range and float32 summary */
    float fl[SY_NTH], wg[SY_NTH];
    unsigned am[SY_NTH];
    int nvv[SY_NTH];
    for (int t = 0; t < SY_NTH; t++) {
      float f = 0.0f;
      unsigned a = 0;
      int nv = 0;
      for (int j = 0; j < SY_V; j++) {
        const int i = SY_V * t + j;
        if (i < Tl) {
          nv++;
          f = fmaf(f, decay_f, alpha * useq[i]);
          if (useq[i] > athr) a |= 1u << j;
        }
      }
      fl[t] = f;
      am[t] = a;
      nvv[t] = nv;
      if (Tl == T) {
        wg[t] = wf[t];
      } else {
        int after = Tl - (SY_V * t + nv);
        wg[t] = (float)dpow[after > 0 ? after : 0];
      }
    }
    /* Where the window energy R has fallen below 1/64 of the largest value the running sums went through since the tile's
     * anchor, float32 has lost it to cancellation: such a tile -- and the one after it, whose first CP means still
     * average this tile's M -- is evaluated in fixed point over its whole length.  (A maximum has no rounding: the
     * running maximum is the same in any order.) */
    int ill = 0;
    {
      float pm = ancc;
      for (int t = 0; t < SY_NTH; t++) {
        pm = fmaxf(pm, rmx[t]);
        if (nvv[t] > 0 && rmn[t] < SY_ILL * pm) ill = 1;
      }
    }
    const int escalate = ill || prev_ill;
    prev_ill = ill;
    float S[4], Pre[4] = {0, 0, 0, 0}, Post[4] = {0, 0, 0, 0};
    int amin_w[4], bmax_w[4];
    for (int w = 0; w < 4; w++) {
      float sv[64];
      for (int l = 0; l < 64; l++) sv[l] = fl[64 * w + l] * wg[64 * w + l];
      sy_scan64(sv);
      S[w] = sv[63];
      amin_w[w] = T;
      bmax_w[w] = -1;
      int l0 = -1, l1 = -1;
      for (int l = 0; l < 64; l++)
        if (am[64 * w + l]) {
          if (l0 < 0) l0 = l;
          l1 = l;
        }
      if (l0 >= 0) {
        const unsigned m0 = am[64 * w + l0], m1 = am[64 * w + l1];
        amin_w[w] = SY_V * (64 * w + l0) + __builtin_ctz(m0);
        bmax_w[w] = SY_V * (64 * w + l1) + 31 - __builtin_clz(m1);
        float pv[64], qv[64];
        for (int l = 0; l < 64; l++) {
          const int t = 64 * w + l;
          float fpre = 0.0f, fpost = 0.0f;
          for (int j = 0; j < SY_V; j++)
            if (j < nvv[t]) {
              const int i = SY_V * t + j;
              fpre = fmaf(fpre, decay_f, alpha * ((i < amin_w[w]) ? useq[i] : 0.0f));
              fpost = fmaf(fpost, decay_f, alpha * ((i > bmax_w[w]) ? useq[i] : 0.0f));
            }
          pv[l] = fpre * wg[t];
          qv[l] = fpost * wg[t];
        }
        sy_scan64(pv);
        sy_scan64(qv);
        Pre[w] = pv[63];
        Post[w] = qv[63];
      }
    }
    int w0 = -1, w1 = -1;
    for (int w = 0; w < 4; w++)
      if (bmax_w[w] >= 0) {
        if (w0 < 0) w0 = w;
        w1 = w;
      }
    if (escalate) {
      tl[g].amin = 0;
      tl[g].bmax = Tl - 1;
      tl[g].gpre = tl[g].gpost = 0.0f;
      int64_t X = 0;
      for (int k = 0; k < Tl; k++) X += sy_q40d((alpha_d * (double)ux[t0 + k]) * dpow[Tl - 1 - k]);
      B[g] = (0.0 + (double)X * QINV) + 0.0;
    } else if (w0 < 0) {
      tl[g].amin = T;
      tl[g].bmax = -1;
      tl[g].gpre = (S[0] + S[1]) + (S[2] + S[3]);
      tl[g].gpost = 0.0f;
      B[g] = (double)tl[g].gpre;
    } else {
      float ga = 0.0f, gb = 0.0f;
      for (int w = 0; w < 4; w++) {
        ga += (w < w0) ? S[w] : (w == w0) ? Pre[w] : 0.0f;
        gb += (w > w1) ? S[w] : (w == w1) ? Post[w] : 0.0f;
      }
      tl[g].amin = amin_w[w0];
      tl[g].bmax = bmax_w[w1];
      tl[g].gpre = ga;
      tl[g].gpost = gb;
      /* the exact part: Q40-rounded products alpha*u*decay^(Tl-1-k), summed as integers */
      int64_t X = 0;
      for (int k = tl[g].amin; k <= tl[g].bmax; k++) X += sy_q40d((alpha_d * (double)ux[t0 + k]) * dpow[Tl - 1 - k]);
      B[g] = ((double)ga + (double)X * QINV) + (double)gb;
    }
    if (range_tap) {
      range_tap[2 * g] = tl[g].bmax >= 0 ? tl[g].amin : -1;
      range_tap[2 * g + 1] = tl[g].bmax;
    }
    if (presel_miss)
      for (int i = 0; i < Tl; i++)
        if (ux[t0 + i] > theta && !(tl[g].bmax >= 0 && i >= tl[g].amin && i <= tl[g].bmax)) (*presel_miss)++;
  }
#undef SY_Y
  /* ---- the average before every tile ---- */
  double *avg_in = (double *)calloc(ntiles ? ntiles : 1, sizeof(double));
  {
    const double A = dpow[T];
    for (uint64_t g = 0; g < ntiles; g++) {
      double acc = 0.0, w = 1.0;
      for (uint64_t k = 1; k <= g; k++) {
        acc += w * B[g - k];
        w *= A;
        if (w < 1e-30) break;
      }
      avg_in[g] = acc;
    }
  }
  /* ---- 4: the state machine on every run of candidates ---- */
  uint64_t g = 0;
  while (g < ntiles) {
    if (tl[g].bmax < 0) {
      g++;
      continue;
    }
    /* runs that START in tile g, in order; a run that reaches the tile's last sample goes on in the next tile */
    const uint64_t t0 = g * (uint64_t)T;
    const int Tl = (t0 + (uint64_t)T <= n) ? T : (int)(n - t0);
    int64_t X = 0;
    int k = tl[g].amin;
    while (k <= tl[g].bmax) {
      const float uk = ux[t0 + (uint64_t)k];
      if (!(uk > theta)) {
        X += sy_q40d((alpha_d * (double)uk) * dpow[Tl - 1 - k]);
        k++;
        continue;
      }
      /* a run starts at k -- unless it continues the previous tile's last run (handled there) */
      int cont = 0;
      if (k == 0 && g > 0 && tl[g - 1].bmax == T - 1 && ux[t0 - 1] > theta) cont = 1;
      if (cont) { /* skip it: walked from the tile it started in */
        while (k <= tl[g].bmax && ux[t0 + (uint64_t)k] > theta) {
          X += sy_q40d((alpha_d * (double)ux[t0 + (uint64_t)k]) * dpow[Tl - 1 - k]);
          k++;
        }
        continue;
      }
      const double bloc = ((double)tl[g].gpre + (double)X * QINV) * (1.0 / dpow[Tl - k]);
      float avg = (float)(avg_in[g] * dpow[k] + bloc);
      int state = 0;
      float peak_val = -INFINITY;
      uint64_t peak_ind = 0;
      uint64_t gg = g;
      int kk = k;
      int open_at_end = 0;
      for (;;) {
        const uint64_t tt0 = gg * (uint64_t)T;
        while (kk <= tl[gg].bmax && ux[tt0 + (uint64_t)kk] > theta) {
          const float u = ux[tt0 + (uint64_t)kk];
          int s1 = state != 0 || (u > avg * rise);
          int newpk = s1 && (u > peak_val);
          if (s1 && !newpk && !(u > avg * fall)) {
            *(uint64_t *)vec_push(peaks, 1) = peak_ind;
            peak_val = -INFINITY;
            s1 = u > avg * rise;
            newpk = s1 && (u > peak_val);
          }
          if (newpk) {
            peak_val = u;
            peak_ind = tt0 + (uint64_t)kk;
          }
          avg = alpha * u + one_m_alpha * avg;
          state = s1;
          kk++;
        }
        /* the run reached the end of the range; does it go on in the next tile? */
        if (kk == T && gg + 1 < ntiles && tl[gg + 1].bmax >= 0 && tl[gg + 1].amin == 0 && ux[tt0 + (uint64_t)T] > theta) {
          gg++;
          kk = 0;
          continue;
        }
        if (tt0 + (uint64_t)kk >= n) open_at_end = 1;
        break;
      }
      if (state == 1 && !open_at_end) *(uint64_t *)vec_push(peaks, 1) = peak_ind;
      /* account for the run's samples inside THIS tile in X, then go on behind it */
      while (k <= tl[g].bmax && ux[t0 + (uint64_t)k] > theta) {
        X += sy_q40d((alpha_d * (double)ux[t0 + (uint64_t)k]) * dpow[Tl - 1 - k]);
        k++;
      }
    }
    g++;
  }
  free(avg_in);
  free(useq);
  free(mt);
  free(mh);
  free(B);
  free(tl);
  free(dpow);
}

typedef struct {
  ofdm_c32 pos[OFDM_MAX_ARITY];
  int arity, nbits, occ, nmap;
  int lanes; /* N/8: threads of the engine's frame workgroup (order of the float32 reductions) */
  int map[OFDM_MAX_FFT];
  ofdm_c32 dfe[OFDM_MAX_FFT];
  float phase, freq, phase_gain, freq_gain, eq_gain;
  int state; /* 0 SYNC_SEARCH, 1 HAVE_SYNC, 2 HAVE_HEADER */
  uint32_t header;
  int hdr_cnt;
  unsigned byte_offset, partial_byte, resid, nresid;
  int packetlen, packetlen_cnt, whitener_offset;
  uint8_t packet[OFDM_MAX_PKT_LEN + 8];
  uint8_t bytes_out[OFDM_MAX_FFT]; /* one symbol never makes more bytes than carriers */
} frame_sink;

static void sink_enter_have_sync(frame_sink *s) {
  s->state = 1;
  s->byte_offset = 0;
  s->partial_byte = 0;
  s->resid = 0;
  s->nresid = 0;
  s->header = 0;
  s->hdr_cnt = 0;
  s->phase = 0.0f;
  s->freq = 0.0f;
  for (int i = 0; i < s->occ; i++) s->dfe[i] = c32(1.0f, 0.0f);
}

static unsigned sink_slicer(const frame_sink *s, ofdm_c32 x) {
  unsigned min_index = 0;
  ofdm_c32 d = c32(x.re - s->pos[0].re, x.im - s->pos[0].im);
  float min_dist = cnorm(d);
  for (int j = 1; j < s->arity; j++) {
    d = c32(x.re - s->pos[j].re, x.im - s->pos[j].im);
    float e = cnorm(d);
    if (e < min_dist) {
      min_dist = e;
      min_index = (unsigned)j;
    }
  }
  return min_index; /* sym_value_out = range(arity), ofdm.py:240 */
}

/* digital_ofdm_frame_sink::demapper */
static unsigned sink_demapper(frame_sink *s, const ofdm_c32 *in, uint8_t *out, ofdm_c32 *derot) {
  unsigned i = 0, bytes_produced = 0;
  ofdm_c32 carrier;
  orc_sincosf(s->phase, &carrier.im, &carrier.re); /* gr_expj(d_phase) */
  /* acc += sigrot * conj(closest): lane t = i mod T of the engine's workgroup sums its carriers in order, the
   * lanes are then added as lane_tree_sum does (normative order of this float32 reduction) */
  int T = s->lanes;
  float pre[OFDM_MAX_FFT / 8], pim[OFDM_MAX_FFT / 8];
  for (int t = 0; t < T; t++) pre[t] = pim[t] = 0.0f;
  unsigned nmap = (unsigned)s->nmap, nb = (unsigned)s->nbits;
  if (derot) memset(derot, 0, sizeof(ofdm_c32) * (size_t)s->occ);
  while (i < nmap) {
    if (s->nresid > 0) {
      s->partial_byte |= s->resid;
      s->byte_offset += s->nresid;
      s->nresid = 0;
      s->resid = 0;
    }
    while (s->byte_offset < 8 && i < nmap) {
      ofdm_c32 sigrot = cmul(cmul(in[s->map[i]], carrier), s->dfe[i]);
      if (derot) derot[i] = sigrot;
      unsigned bits = sink_slicer(s, sigrot);
      ofdm_c32 closest = s->pos[bits];
      ofdm_c32 e = cmul_conj(sigrot, closest);
      pre[i % (unsigned)T] = pre[i % (unsigned)T] + e.re;
      pim[i % (unsigned)T] = pim[i % (unsigned)T] + e.im;
      float sden = cnorm(sigrot);
      if (sden > 0.001f) {
        /* pos[bits] / sigrot as the conjugate product times ONE reciprocal (normative; std::complex's own division is
         * not reproducible across libraries either) */
        float sinv = 1.0f / sden;
        ofdm_c32 q = c32((closest.re * sigrot.re + closest.im * sigrot.im) * sinv,
                         (closest.im * sigrot.re - closest.re * sigrot.im) * sinv);
        s->dfe[i].re = s->dfe[i].re + s->eq_gain * (q.re - s->dfe[i].re);
        s->dfe[i].im = s->dfe[i].im + s->eq_gain * (q.im - s->dfe[i].im);
      }
      i++;
      if (8 - s->byte_offset >= nb) {
        s->partial_byte |= bits << s->byte_offset;
        s->byte_offset += nb;
      } else {
        s->nresid = nb - (8 - s->byte_offset);
        unsigned mask = (1u << (8 - s->byte_offset)) - 1u;
        s->partial_byte |= (bits & mask) << s->byte_offset;
        s->resid = bits >> (8 - s->byte_offset);
        s->byte_offset += nb - s->nresid;
      }
    }
    if (s->byte_offset == 8) {
      out[bytes_produced++] = (uint8_t)s->partial_byte;
      s->byte_offset = 0;
      s->partial_byte = 0;
    }
  }
  float angle = orc_atan2f(lane_tree_sum(pim, T), lane_tree_sum(pre, T)); /* arg(acc), bit-reproducible form */
  s->freq = s->freq - s->freq_gain * angle;
  s->phase = s->phase + s->freq - s->phase_gain * angle;
  if (s->phase >= 6.28318530717958647692f) s->phase -= 6.28318530717958647692f;
  if (s->phase < 0.0f) s->phase += 6.28318530717958647692f;
  return bytes_produced;
}

static void rx_post_message(orc_rx_result *r, const ofdm_cfg *cfg, const uint8_t *msg, int len) {
  *(uint64_t *)vec_push(&r->raw_off, 1) = r->raw.n;
  memcpy(vec_push(&r->raw, (size_t)len), msg, (size_t)len);
  /* _queue_watcher_thread.run: unmake_packet(msg.to_string()) (ofdm.py:300-305) */
  uint8_t pay[OFDM_MASK_LEN];
  uint32_t plen = 0;
  int ok = 0;
  orc_unmake_packet(cfg, msg, (uint32_t)len, pay, &plen, &ok);
  *(uint64_t *)vec_push(&r->pay_off, 1) = r->payload.n;
  memcpy(vec_push(&r->payload, plen), pay, plen);
  *(uint32_t *)vec_push(&r->pay_len, 1) = plen;
  *(uint8_t *)vec_push(&r->pay_ok, 1) = (uint8_t)ok;
  r->st.packets++;
  if (ok) r->st.crc_ok++;
}

#define YAT(ix) (((ix) >= 0 && (ix) < N) ? Y[(ix)] : c32(0.0f, 0.0f))

orc_rx_result *orc_rx(const ofdm_cfg *cfg, const ofdm_c32 *iq, uint64_t n, uint32_t tap_mask) {
  int N = (int)cfg->fft_length, CP = (int)cfg->cp_length, occ = (int)cfg->occupied_tones;
  int L = N + CP;
  orc_rx_result *r = (orc_rx_result *)calloc(1, sizeof(*r));
  r->tap_mask = tap_mask;
  r->N = N;
  r->occ = occ;
  vec_init(&r->y, sizeof(ofdm_c32));
  vec_init(&r->metric, sizeof(float));
  vec_init(&r->peaks, sizeof(uint64_t));
  vec_init(&r->peaks_gr, sizeof(uint64_t));
  vec_init(&r->presel, sizeof(float));
  vec_init(&r->ranges, sizeof(int32_t));
  vec_init(&r->angles, sizeof(float));
  vec_init(&r->frames, sizeof(uint64_t));
  vec_init(&r->fft, sizeof(ofdm_c32));
  vec_init(&r->acq, sizeof(ofdm_c32));
  vec_init(&r->sink, sizeof(ofdm_c32));
  vec_init(&r->sampler, sizeof(ofdm_c32));
  vec_init(&r->sigmix, sizeof(ofdm_c32));
  vec_init(&r->nco, sizeof(ofdm_c32));
  vec_init(&r->raw, 1);
  vec_init(&r->raw_off, sizeof(uint64_t));
  vec_init(&r->payload, 1);
  vec_init(&r->pay_off, sizeof(uint64_t));
  vec_init(&r->pay_len, sizeof(uint32_t));
  vec_init(&r->pay_ok, 1);
  r->st.samples = n;
  if (n == 0) {
    *(uint64_t *)vec_push(&r->raw_off, 1) = 0;
    *(uint64_t *)vec_push(&r->pay_off, 1) = 0;
    return r;
  }

  int fixed = cfg->sync_mode == OFDM_SYNC_FIXED;
  ofdm_c32 *y = (ofdm_c32 *)vec_push(&r->y, n);
  ofdm_c32 *P = (ofdm_c32 *)calloc(n, sizeof(ofdm_c32));
  if (!fixed) {
    /* --- chan_filt ------------------------------------------------------- */
    chan_filter(cfg, iq, n, y);
    /* --- ofdm_sync_pn ---------------------------------------------------- */
    float *u = (float *)vec_push(&r->metric, n);
    sync_metric(cfg, y, n, u, P);
    float *u32_tap = (tap_mask & (1u << OFDM_TAP_RX_PRESEL)) ? (float *)vec_push(&r->presel, n) : NULL;
    int32_t *range_tap = (int32_t *)vec_push(&r->ranges, 2 * (size_t)((n + SY_T - 1) / SY_T));
    peak_detect(cfg, y, u, n, &r->peaks, u32_tap, range_tap, &r->presel_miss);
    peak_detect_gr(cfg, u, n, &r->peaks_gr);
  } else {
    /* SYNC = "fixed" (ofdm_receiver.py~:108-119, "for testing only"): chan_filt = gr.multiply_const_cc(1.0);
     * ofdm_sync_fixed: a vector source repeating nsymbols*(N+CP) bytes with a 1 at index (N+CP)-1, and a constant
     * frequency-offset stream into the NCO. */
    memcpy(y, iq, sizeof(ofdm_c32) * n);
    uint64_t period = (uint64_t)cfg->fixed_nsymbols * (uint64_t)L;
    for (uint64_t p0 = (uint64_t)L - 1; p0 < n; p0 += period) *(uint64_t *)vec_push(&r->peaks, 1) = p0;
  }
  uint64_t npk = r->peaks.n;
  const uint64_t *pk = (const uint64_t *)r->peaks.p;
  r->st.peaks = npk;

  /* gr_sample_and_hold_ff(complex_to_arg(P), timing) + gr_frequency_modulator_fc(-2/N)
   * (ofdm_receiver.py~:98,123,133): closed form, float64 phase.  Phi[j] = phase of the
   * last sample before flag j takes effect; for pk[j] <= n < pk[j+1]:
   * phi[n] = Phi[j] + step[j]*(n - pk[j] + 1).                                   */
  float sens = (float)(-2.0 / (double)N);
  float *ang = (float *)vec_push(&r->angles, npk);
  double *Phi = (double *)malloc(sizeof(double) * (npk + 1));
  double *step = (double *)malloc(sizeof(double) * (npk + 1));
  /* The phase a flag starts from is kept modulo one turn as an integer (unit 2^-64 turn): integer addition is
   * associative, so the engine's parallel scan over the flags gives the same bits as this running sum. */
  uint64_t phi_u = 0;
  /* the NCO line before the first flag: phase 0 -- except SYNC "fixed", whose constant frequency input drives the
   * NCO from the first sample: phi[n] = ref_step * (n + 1) */
  int ref_on = fixed && cfg->fixed_freq_offset != 0.0f;
  double ref_step = (double)(sens * cfg->fixed_freq_offset);
  if (ref_on && npk > 0) {
    double t = ref_step * (double)pk[0] * 0.15915494309189533577;
    t -= floor(t);
    phi_u = (uint64_t)(t * 18446744073709551616.0);
  }
  for (uint64_t j = 0; j < npk; j++) {
    ang[j] = fixed ? cfg->fixed_freq_offset : orc_atan2f(P[pk[j]].im, P[pk[j]].re);
    step[j] = (double)(sens * ang[j]);
    Phi[j] = (double)(int64_t)phi_u * 3.4061215800865545e-19; /* 2 pi / 2^64: phase in [-pi, pi) */
    if (j + 1 < npk) {
      double t = step[j] * (double)(pk[j + 1] - pk[j]) * 0.15915494309189533577; /* turns */
      t -= floor(t);
      phi_u += (uint64_t)(t * 18446744073709551616.0);
    }
  }
  free(P);
  /* ofdm_receiver-sigmix_c.dat / -nco_c.dat: the closed form sample by sample over the whole stream */
  if (tap_mask & ((1u << OFDM_TAP_RX_SIGMIX) | (1u << OFDM_TAP_RX_NCO))) {
    ofdm_c32 *sm = (tap_mask & (1u << OFDM_TAP_RX_SIGMIX)) ? (ofdm_c32 *)vec_push(&r->sigmix, n) : NULL;
    ofdm_c32 *nc = (tap_mask & (1u << OFDM_TAP_RX_NCO)) ? (ofdm_c32 *)vec_push(&r->nco, n) : NULL;
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; i++) {
      while (cnt < npk && pk[cnt] <= i) cnt++;
      double ph = 0.0;
      if (cnt > 0) ph = Phi[cnt - 1] + step[cnt - 1] * (double)(i - pk[cnt - 1] + 1);
      else if (ref_on) ph = 0.0 + ref_step * (double)((int64_t)i - 0 + 1);
      dcx rr = orc_expj(ph);
      ofdm_c32 rot = c32((float)rr.re, (float)rr.im);
      if (nc) nc[i] = rot;
      if (sm) sm[i] = cmul(y[i], rot);
    }
  }

  /* --- per-symbol machinery -------------------------------------------- */
  int zl = (N - occ + 1) / 2;
  int shift = (int)cfg->max_fft_shift_len;
  fft_plan plan;
  fft_plan_init(&plan, N);
  ofdm_c32 *win = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)N);
  ofdm_c32 *Y = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)N);
  ofdm_c32 *acq = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)occ);
  ofdm_c32 *hinv = (ofdm_c32 *)calloc((size_t)occ, sizeof(ofdm_c32));
  float *kd = (float *)calloc((size_t)occ, sizeof(float));
  float *sd = (float *)calloc((size_t)N, sizeof(float));
  ofdm_c32 *derot = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)occ);
  /* digital_ofdm_frame_acquisition ctor: known_phase_diff */
  for (int i = 0; i + 2 < occ; i += 2) {
    ofdm_c32 a = cfg->known_symbol[i], b = cfg->known_symbol[i + 2];
    kd[i] = cnorm(c32(a.re - b.re, a.im - b.im));
  }
  int coarse = 0;
  unsigned phase_count = 1;

  frame_sink *sk = (frame_sink *)calloc(1, sizeof(frame_sink));
  sk->arity = (int)cfg->arity;
  sk->nbits = orc_nbits(cfg);
  sk->occ = occ;
  sk->lanes = N / 8;
  memcpy(sk->pos, cfg->constellation, sizeof(ofdm_c32) * cfg->arity);
  sk->nmap = orc_carrier_map2(occ, occ, cfg->carrier_map, 1, sk->map, OFDM_MAX_FFT);
  sk->phase_gain = cfg->phase_gain;
  sk->freq_gain = cfg->freq_gain;
  sk->eq_gain = cfg->eq_gain;
  for (int i = 0; i < occ; i++) sk->dfe[i] = c32(1.0f, 0.0f);
  sk->state = 0;
  if (sk->nmap < 0) sk->nmap = 0;

  /* --- digital_ofdm_sampler(N, N+CP, timeout) automaton (ofdm_receiver.py~:125) --- */
  enum { ST_NO_SIG, ST_PREAMBLE, ST_FRAME };
  int sstate = ST_NO_SIG;
  int64_t timeout = 0;
  uint64_t base = 0;
  uint64_t next_pk = 0;     /* first peak with index >= scan start */
  uint64_t cur_frame_data = 0;
  /* NCO recurrence state of the current frame */
  dcx *nco_base = (dcx *)malloc(sizeof(dcx) * (size_t)(N / 8));
  dcx nco_RL = {1.0, 0.0}, nco_RT = {1.0, 0.0};
  uint64_t nco_j = 0;
  while (base + (uint64_t)L + (uint64_t)N < n) { /* the scan touches trigger[base+L+N] */
    /* search trigger[base+N .. base+L+N] unless already in PREAMBLE */
    uint64_t lo = base + (uint64_t)N, hi = base + (uint64_t)L + (uint64_t)N;
    while (next_pk < npk && pk[next_pk] < lo) next_pk++;
    int found = (next_pk < npk && pk[next_pk] <= hi);
    uint64_t sym_start;
    int flag;
    if (found) {
      uint64_t p = pk[next_pk];
      sstate = ST_PREAMBLE;
      sym_start = p - (uint64_t)N + 1;
      flag = 1;
      timeout = (int64_t)cfg->sampler_timeout;
      sstate = ST_FRAME;
      base = sym_start; /* consume_each(index - N + 1) */
      next_pk++;
      uint64_t *fr = (uint64_t *)vec_push(&r->frames, 2);
      fr[0] = p;
      fr[1] = 0;
      cur_frame_data = r->frames.n - 1;
      r->st.frames++;
    } else if (sstate == ST_FRAME) {
      sym_start = base + (uint64_t)L;
      flag = 0;
      if (timeout-- == 0) sstate = ST_NO_SIG;
      base += (uint64_t)L;
      ((uint64_t *)r->frames.p)[cur_frame_data]++;
    } else {
      base += (uint64_t)L + 1; /* consume_each(index - N), index ran to L+N+1 */
      continue;
    }
    r->st.symbols++;

    /* --- sigmix: chan_filt * nco over this symbol's N samples ---------------
     * The NCO phasor exp(j phi[n]) is evaluated by recurrence in float64 (GR's own NCO is a recurrence too), in the
     * normative order the engine uses: the symbol's sample i = t + m*T (T = N/8, t < T, m < 8) gets
     *   data symbol k >= 1 of the frame of flag j:  base_t(k) * RT^m,  base_t(k) = base_t(k-1) * RL,
     *       base_t(0) = expj(Phi[j] + step[j]*(2 - N + t)),  RL = expj(step[j]*L),  RT = expj(step[j]*T),
     *   the preamble symbol (it ENDS on the flag, so it runs on the previous flag's line except for its
     *       last sample):  Ap_t * RTp^m,  Ap_t = expj(Phi[j-1] + step[j-1]*(n - pk[j-1] + 1)) at n = first
     *       sample + t, RTp = expj(step[j-1]*T); sample N-1 = expj(Phi[j] + step[j]),
     *   products taken left to right, rounded to float32 at the end; if the previous flag lies inside the preamble
     *   symbol every sample evaluates its own closed form.                                                   */
    {
      int T = N / 8;
      if (flag) {
        uint64_t j = next_pk - 1; /* the flag that started this frame */
        double st = step[j];
        nco_j = j;
        nco_RL = orc_expj(st * (double)L);
        nco_RT = orc_expj(st * (double)T);
        for (int t = 0; t < T; t++) nco_base[t] = orc_expj(Phi[j] + st * (double)(2 - N + t));
        int pre_simple = (j == 0) || (pk[j - 1] <= sym_start);
        if (pre_simple) {
          dcx RTp = {1.0, 0.0};
          double stq = 0.0;
          if (j > 0) {
            stq = step[j - 1];
            RTp = orc_expj(stq * (double)T);
          }
          dcx Rflag = orc_expj(Phi[j] + st);
          if (j == 0 && ref_on) RTp = orc_expj(ref_step * (double)T);
          for (int t = 0; t < T; t++) {
            dcx rr = {1.0, 0.0};
            if (j > 0) rr = orc_expj(Phi[j - 1] + stq * (double)((int64_t)(sym_start + (uint64_t)t) - (int64_t)pk[j - 1] + 1));
            else if (ref_on) rr = orc_expj(0.0 + ref_step * (double)((int64_t)(sym_start + (uint64_t)t) - 0 + 1));
            for (int m = 0; m < 8; m++) {
              int i = t + m * T;
              ofdm_c32 rot = c32((float)rr.re, (float)rr.im);
              if (i == N - 1) rot = c32((float)Rflag.re, (float)Rflag.im);
              win[i] = cmul(y[sym_start + (uint64_t)i], rot);
              rr = dmul(rr, RTp);
            }
          }
        } else {
          for (int i = 0; i < N; i++) {
            uint64_t idx = sym_start + (uint64_t)i;
            int64_t q = (int64_t)j;
            while (q >= 0 && pk[q] > idx) q--;
            double ph = 0.0;
            if (q >= 0) ph = Phi[q] + step[q] * (double)(idx - pk[q] + 1);
            else if (ref_on) ph = 0.0 + ref_step * (double)((int64_t)idx - 0 + 1);
            dcx rr = orc_expj(ph);
            win[i] = cmul(y[idx], c32((float)rr.re, (float)rr.im));
          }
        }
      } else {
        (void)nco_j;
        for (int t = 0; t < T; t++) {
          nco_base[t] = dmul(nco_base[t], nco_RL);
          dcx rr = nco_base[t];
          for (int m = 0; m < 8; m++) {
            int i = t + m * T;
            win[i] = cmul(y[sym_start + (uint64_t)i], c32((float)rr.re, (float)rr.im));
            rr = dmul(rr, nco_RT);
          }
        }
      }
    }

    if (tap_mask & (1u << OFDM_TAP_RX_SAMPLER)) memcpy(vec_push(&r->sampler, (size_t)N), win, sizeof(ofdm_c32) * (size_t)N);
    /* --- gr.fft_vcc(N, True, [1]*N, True): forward DFT, output halves swapped -- */
    fft_exec(&plan, win, 0);
    for (int k = 0; k < N; k++) Y[k] = win[(k + N / 2) % N];
    if (tap_mask & (1u << OFDM_TAP_RX_FFT)) memcpy(vec_push(&r->fft, (size_t)N), Y, sizeof(ofdm_c32) * (size_t)N);

    /* --- digital_ofdm_frame_acquisition(occ, N, CP, ks[0], 4) ---------------- */
    if (flag) {
      phase_count = 1;
      /* correlate() */
      for (int i = 0; i < N; i++) sd[i] = 0.0f;
      for (int i = 0; i < N - 2; i++) sd[i] = cnorm(c32(Y[i].re - Y[i + 2].re, Y[i].im - Y[i + 2].im));
      int index = 0;
      float mx = 0.0f;
      for (int i = zl - shift; i < zl + shift; i++) {
        /* sum_j kd[j] * sd[i+j]: lane t = j mod T sums its terms in order, lanes added by lane_tree_sum */
        int T = N / 8;
        float part[OFDM_MAX_FFT / 8];
        for (int t = 0; t < T; t++) part[t] = 0.0f;
        for (int j = 0; j < occ; j++) {
          int q = i + j;
          float s2 = (q >= 0 && q < N) ? sd[q] : 0.0f;
          part[j % T] = part[j % T] + kd[j] * s2;
        }
        float sum = lane_tree_sum(part, T);
        if (sum > mx) {
          mx = sum;
          index = i;
        }
      }
      coarse = index - zl;
      /* calculate_equalizer() */
      {
        double a = -2.0 * M_PI * (double)coarse * (double)CP / (double)N * 1.0;
        float af = (float)a;
        ofdm_c32 comp;
        orc_sincosf(af, &comp.im, &comp.re); /* gr_expj(af) */
        hinv[0] = cdiv(cfg->known_symbol[0], cmul(comp, YAT(zl + coarse)));
        for (int i = 2; i < occ; i += 2) {
          hinv[i] = cdiv(cfg->known_symbol[i], cmul(comp, YAT(i + zl + coarse)));
          hinv[i - 1] = c32((hinv[i].re + hinv[i - 2].re) / 2.0f, (hinv[i].im + hinv[i - 2].im) / 2.0f);
        }
        if (!(occ & 1)) hinv[occ - 1] = hinv[occ - 2];
      }
    }
    {
      double a = -2.0 * M_PI * (double)coarse * (double)CP / (double)N * (double)phase_count;
      float af = (float)a;
      ofdm_c32 comp;
      orc_sincosf(af, &comp.im, &comp.re);
      for (int i = 0; i < occ; i++) acq[i] = cmul(cmul(hinv[i], comp), YAT(i + zl + coarse));
      phase_count++;
      if (phase_count == 1000) phase_count = 1; /* MAX_NUM_SYMBOLS */
    }
    if (tap_mask & (1u << OFDM_TAP_RX_ACQ)) memcpy(vec_push(&r->acq, (size_t)occ), acq, sizeof(ofdm_c32) * (size_t)occ);

    /* --- digital_ofdm_frame_sink::work (ofdm.py:240-247) --------------------- */
    if (sk->state == 0) {
      if (flag) sink_enter_have_sync(sk);
    } else if (sk->state == 1) {
      unsigned bytes = sink_demapper(sk, acq, sk->bytes_out, derot);
      if (flag) r->st.chained_frames++;
      if (tap_mask & (1u << OFDM_TAP_RX_SINK)) memcpy(vec_push(&r->sink, (size_t)occ), derot, sizeof(ofdm_c32) * (size_t)occ);
      unsigned j = 0;
      while (j < bytes) {
        sk->header = (sk->header << 8) | (uint32_t)sk->bytes_out[j];
        j++;
        if (++sk->hdr_cnt == 4) {
          if (((sk->header >> 16) ^ (sk->header & 0xFFFF)) == 0) {
            r->st.headers_ok++;
            sk->state = 2;
            sk->packetlen = (int)((sk->header >> 16) & 0x0FFF);
            sk->whitener_offset = (int)((sk->header >> 28) & 0xF);
            sk->packetlen_cnt = 0;
            while (j < bytes && sk->packetlen_cnt < sk->packetlen) sk->packet[sk->packetlen_cnt++] = sk->bytes_out[j++];
            if (sk->packetlen_cnt == sk->packetlen) {
              rx_post_message(r, cfg, sk->packet, sk->packetlen);
              sk->state = 0;
            }
          } else {
            sk->state = 0; /* bad header */
          }
        }
      }
    } else {
      unsigned bytes = sink_demapper(sk, acq, sk->bytes_out, derot);
      if (flag) r->st.chained_frames++;
      if (tap_mask & (1u << OFDM_TAP_RX_SINK)) memcpy(vec_push(&r->sink, (size_t)occ), derot, sizeof(ofdm_c32) * (size_t)occ);
      unsigned j = 0;
      while (j < bytes) {
        sk->packet[sk->packetlen_cnt++] = sk->bytes_out[j++];
        if (sk->packetlen_cnt == sk->packetlen) {
          rx_post_message(r, cfg, sk->packet, sk->packetlen);
          sk->state = 0;
          break;
        }
      }
    }
  }
  *(uint64_t *)vec_push(&r->raw_off, 1) = r->raw.n;
  *(uint64_t *)vec_push(&r->pay_off, 1) = r->payload.n;

  free(nco_base);
  free(sk);
  free(derot);
  free(sd);
  free(kd);
  free(hinv);
  free(acq);
  free(Y);
  free(win);
  fft_plan_free(&plan);
  free(step);
  free(Phi);
  if (!(tap_mask & (1u << OFDM_TAP_RX_CHAN_FILT))) vec_free(&r->y);
  if (!(tap_mask & (1u << OFDM_TAP_RX_METRIC))) vec_free(&r->metric);
  return r;
}

uint64_t orc_rx_tap(const orc_rx_result *r, int tap, void *out, uint64_t cap_bytes) {
  const vec *v = NULL;
  switch (tap) {
    case OFDM_TAP_RX_CHAN_FILT: v = &r->y; break;
    case OFDM_TAP_RX_METRIC: v = &r->metric; break;
    case OFDM_TAP_RX_PEAKS: v = &r->peaks; break;
    case ORC_TAP_PEAKS_GR: v = &r->peaks_gr; break;
    case OFDM_TAP_RX_PRESEL: v = &r->presel; break;
    case ORC_TAP_RANGES: v = &r->ranges; break;
    case OFDM_TAP_RX_ANGLES: v = &r->angles; break;
    case OFDM_TAP_RX_FRAMES: v = &r->frames; break;
    case OFDM_TAP_RX_FFT: v = &r->fft; break;
    case OFDM_TAP_RX_ACQ: v = &r->acq; break;
    case OFDM_TAP_RX_SINK: v = &r->sink; break;
    case OFDM_TAP_RX_SAMPLER: v = &r->sampler; break;
    case OFDM_TAP_RX_SIGMIX: v = &r->sigmix; break;
    case OFDM_TAP_RX_NCO: v = &r->nco; break;
    case OFDM_TAP_RX_PACKETS: v = &r->raw; break;
    default: return 0;
  }
  uint64_t nb = (uint64_t)v->n * v->esz;
  if (out && cap_bytes >= nb && nb) memcpy(out, v->p, nb);
  return nb;
}

int orc_rx_npackets(const orc_rx_result *r) { return (int)r->pay_len.n; }
uint64_t orc_rx_payload_bytes(const orc_rx_result *r) { return r->payload.n; }

int orc_rx_packets(const orc_rx_result *r, uint8_t *payload_out, uint64_t cap, uint64_t *off, uint32_t *len,
                   uint8_t *ok, int max_pkts) {
  int np = (int)r->pay_len.n;
  if (np > max_pkts || r->payload.n > cap) return OFDM_E_CAPACITY;
  if (r->payload.n) memcpy(payload_out, r->payload.p, r->payload.n);
  memcpy(off, r->pay_off.p, sizeof(uint64_t) * (size_t)(np + 1));
  if (np) {
    memcpy(len, r->pay_len.p, sizeof(uint32_t) * (size_t)np);
    memcpy(ok, r->pay_ok.p, (size_t)np);
  }
  return np;
}

void orc_rx_stats(const orc_rx_result *r, ofdm_stats *st) { *st = r->st; }
uint64_t orc_rx_presel_miss(const orc_rx_result *r) { return r->presel_miss; }

void orc_rx_free(orc_rx_result *r) {
  if (!r) return;
  vec_free(&r->y);
  vec_free(&r->metric);
  vec_free(&r->peaks);
  vec_free(&r->peaks_gr);
  vec_free(&r->presel);
  vec_free(&r->ranges);
  vec_free(&r->angles);
  vec_free(&r->frames);
  vec_free(&r->fft);
  vec_free(&r->acq);
  vec_free(&r->sink);
  vec_free(&r->sampler);
  vec_free(&r->sigmix);
  vec_free(&r->nco);
  vec_free(&r->raw);
  vec_free(&r->raw_off);
  vec_free(&r->payload);
  vec_free(&r->pay_off);
  vec_free(&r->pay_len);
  vec_free(&r->pay_ok);
  free(r);
}

/* ------------------------------------------------------------------------ */
/* Spectrum sensor: predictive_sense.py `sensor` graph (:72-123) + sense_loop
 * (:150-222) + hex_conv (:235-268).  The GNU Radio blocks behind it
 * (gr_stream_to_vector, gr_fft_vcc with a window, gr_complex_to_mag_squared,
 * gr_bin_statistics_f) are absent from the reference tree: restated from
 * GNU Radio 3.6.0 [SURVEY A.13] -- PARITY UNPINNED for the message bodies.
 * sense_loop's tail (mean of 10, threshold, half swap, hex) is in the tree and
 * PINNED by the 43 recorded blocks of output.txt / output_with_detection.txt /
 * crap.txt (tests/golden/sense_blocks.json) and final_hex_conv.py:37.        */
/* ------------------------------------------------------------------------ */
static int sense_cfg_ok(const ofdm_sense_cfg *sc) {
  if (!sc || sc->struct_size != sizeof(ofdm_sense_cfg)) return 0;
  uint32_t n = sc->fft_size;
  if (n < 64 || n > OFDM_SENSE_MAX_FFT || (n & (n - 1))) return 0;
  if (sc->dwell_delay < 1 || sc->avg_msgs < 1) return 0;
  return 1;
}

int orc_sense_count(const ofdm_sense_cfg *sc, uint64_t nsamples, uint64_t *nmsgs, uint64_t *ndecisions) {
  if (!sense_cfg_ok(sc)) return OFDM_E_INVAL;
  /* gr_bin_statistics_f::work: ST_TUNE_DELAY eats tune_delay vectors, ST_DWELL_DELAY
   * accrues dwell_delay vectors, then send_stats() and back to ST_TUNE_DELAY */
  uint64_t nvec = nsamples / sc->fft_size;
  uint64_t period = (uint64_t)sc->tune_delay + sc->dwell_delay;
  uint64_t nm = nvec / period;
  if (nmsgs) *nmsgs = nm;
  /* sense_loop: avg_msgs messages are summed; the next one lands in the else branch
   * (:174) where the decision is taken and its own data is discarded */
  if (ndecisions) *ndecisions = nm / ((uint64_t)sc->avg_msgs + sc->skip_msgs);
  return OFDM_OK;
}

int orc_sense_decide(const ofdm_sense_cfg *sc, const float *msgs, uint64_t nmsgs, double *mean_inorder,
                     uint8_t *bits_inorder, char *hex) {
  if (!sense_cfg_ok(sc)) return OFDM_E_INVAL;
  static const char abc[] = "0123456789ABCDEF"; /* hex_conv :239 */
  int S = (int)sc->fft_size, H = S / 2;
  uint64_t per = (uint64_t)sc->avg_msgs + sc->skip_msgs;
  uint64_t nd = nmsgs / per;
  double *acc = (double *)malloc(sizeof(double) * (size_t)S);
  uint8_t *thr = (uint8_t *)malloc((size_t)S);
  uint8_t *ino = (uint8_t *)malloc((size_t)S);
  for (uint64_t d = 0; d < nd; d++) {
    for (int i = 0; i < S; i++) acc[i] = 0.0; /* moving_avg_data = [0]*size */
    for (uint32_t k = 0; k < sc->avg_msgs; k++) {
      const float *m = msgs + (d * per + k) * (uint64_t)S;
      for (int i = 0; i < S; i++) acc[i] = acc[i] + (double)m[i]; /* :171, struct.unpack('f') -> float64 */
    }
    for (int i = 0; i < S; i++) {
      acc[i] = acc[i] / (double)sc->avg_msgs;   /* :176 */
      thr[i] = acc[i] > sc->threshold ? 0 : 1;  /* :179 */
    }
    for (int i = 0; i < H; i++) { /* :193-205: upper half of the FFT output first */
      ino[i] = thr[i + H];
      ino[i + H] = thr[i];
      if (mean_inorder) {
        mean_inorder[d * (uint64_t)S + i] = acc[i + H];
        mean_inorder[d * (uint64_t)S + i + H] = acc[i];
      }
    }
    if (bits_inorder) memcpy(bits_inorder + d * (uint64_t)S, ino, (size_t)S);
    if (hex) { /* hex_conv: 4 bits per character, first bit = 2**0 (:243-248) */
      for (int j = 0; j + 4 <= S; j += 4) {
        int v = ino[j] | (ino[j + 1] << 1) | (ino[j + 2] << 2) | (ino[j + 3] << 3);
        hex[d * (uint64_t)(S / 4) + j / 4] = abc[v];
      }
    }
  }
  free(acc);
  free(thr);
  free(ino);
  return OFDM_OK;
}

int orc_sense(const ofdm_sense_cfg *sc, const ofdm_c32 *iq, uint64_t nsamples, float *msgs, double *mean_inorder,
              uint8_t *bits_inorder, char *hex) {
  uint64_t nm = 0, nd = 0;
  int rc = orc_sense_count(sc, nsamples, &nm, &nd);
  if (rc != OFDM_OK) return rc;
  int S = (int)sc->fft_size;
  uint64_t period = (uint64_t)sc->tune_delay + sc->dwell_delay;
  float *mbuf = msgs ? msgs : (float *)malloc(sizeof(float) * (size_t)(nm ? nm : 1) * (size_t)S);
  ofdm_c32 *v = (ofdm_c32 *)malloc(sizeof(ofdm_c32) * (size_t)S);
  fft_plan plan;
  fft_plan_init(&plan, S);
  for (uint64_t m = 0; m < nm; m++) {
    float *mx = mbuf + m * (uint64_t)S;
    for (int i = 0; i < S; i++) mx[i] = 0.0f; /* reset_stats(): d_max = 0 */
    for (uint64_t f = m * period + sc->tune_delay; f < (m + 1) * period; f++) {
      const ofdm_c32 *x = iq + f * (uint64_t)S;
      /* gr_fft_vcc_fftw::work, forward with window: dst[i] = in[i] * window[i]; no shift */
      for (int i = 0; i < S; i++) v[i] = c32(x[i].re * sc->window[i], x[i].im * sc->window[i]);
      fft_exec(&plan, v, 0);
      for (int i = 0; i < S; i++) {
        float p = v[i].re * v[i].re + v[i].im * v[i].im; /* gr_complex_to_mag_squared */
        if (p > mx[i]) mx[i] = p;                          /* accrue_stats: max */
      }
    }
  }
  fft_plan_free(&plan);
  free(v);
  rc = orc_sense_decide(sc, mbuf, nm, mean_inorder, bits_inorder, hex);
  if (!msgs) free(mbuf);
  return rc;
}
