/*
 * ofdm_oracle.h -- CPU oracle for the OFDM hot path.  TEST INFRASTRUCTURE ONLY
 * (parity checker + "port" CPU baseline); see the header of ofdm_oracle.c.
 * Shares the POD configuration struct with the product ABI so that one Python
 * ctypes structure drives both sides.
 */
#ifndef OFDM_ORACLE_H
#define OFDM_ORACLE_H
#include <stdint.h>
#include "../include/ofdm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_rx_result orc_rx_result;

int orc_nbits(const ofdm_cfg *cfg);
float orc_atan2f(float y, float x);
void orc_sincosf(float x, float *sn, float *cs);
void orc_sincosf_vec(const float *x, uint64_t n, float *sn, float *cs);
void orc_expj_vec(const double *ph, uint64_t n, double *re, double *im);
int orc_fft(ofdm_c32 *x, int n, int inverse);
int orc_filter_fft_len(int ntaps);
uint32_t orc_crc32(const uint8_t *buf, uint64_t len);
int orc_framed_len(const ofdm_cfg *cfg, uint32_t payload_len, uint32_t *out);
int orc_make_packet(const ofdm_cfg *cfg, const uint8_t *payload, uint32_t len, uint8_t *out, uint32_t *outlen);
int orc_unmake_packet(const ofdm_cfg *cfg, const uint8_t *msg, uint32_t len, uint8_t *payload_out,
                      uint32_t *payload_len, int *ok);
int orc_carrier_map(int occ, int container, const char *carriers, int *map, int cap);
int orc_carrier_map2(int occ, int container, const char *carriers, int sink, int *map, int cap);
uint32_t orc_tx_data_symbols(const ofdm_cfg *cfg, uint32_t framed_len, int ncarriers);
uint32_t orc_pad_symbol(uint64_t seed, uint64_t pkt, uint64_t slot, uint32_t arity);
int orc_tx(const ofdm_cfg *cfg, const uint8_t *payloads, const uint64_t *payload_off, const uint32_t *payload_len,
           int npkt, uint64_t lead, ofdm_c32 *iq_out, uint64_t iq_cap, uint64_t *nsamples_out, ofdm_c32 *freq_tap,
           uint8_t *framed_tap, uint64_t *framed_off_tap);
int orc_tx_ex(const ofdm_cfg *cfg, const uint8_t *payloads, const uint64_t *payload_off, const uint32_t *payload_len,
              int npkt, uint64_t lead, ofdm_c32 *iq_out, uint64_t iq_cap, uint64_t *nsamples_out, ofdm_c32 *freq_tap,
              uint8_t *framed_tap, uint64_t *framed_off_tap, ofdm_c32 *ifft_tap);
void orc_philox(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t out[2]);
void orc_philox_r(uint64_t seed, uint64_t stream, uint64_t idx, int rounds, uint32_t out[2]);
int orc_channel(ofdm_c32 *iq, uint64_t n, const ofdm_chan *ch, uint64_t index0);
orc_rx_result *orc_rx(const ofdm_cfg *cfg, const ofdm_c32 *iq, uint64_t n, uint32_t tap_mask);
uint64_t orc_rx_tap(const orc_rx_result *r, int tap, void *out, uint64_t cap_bytes);
int orc_rx_npackets(const orc_rx_result *r);
uint64_t orc_rx_payload_bytes(const orc_rx_result *r);
int orc_rx_packets(const orc_rx_result *r, uint8_t *payload_out, uint64_t cap, uint64_t *off, uint32_t *len,
                   uint8_t *ok, int max_pkts);
void orc_rx_stats(const orc_rx_result *r, ofdm_stats *st);
/* oracle-only taps (orc_rx_tap): the flags of gr_peak_detector_fb run LITERALLY (float32 recurrence of the average from
 * the first sample) -- the cross-check of the normative evaluation; the exact-evaluation range of every tile */
#define ORC_TAP_PEAKS_GR 100 /* u64[] */
#define ORC_TAP_RANGES 102   /* i32[ntiles][2] */
/* samples whose exact metric is above the candidate threshold but which the float32 pre-selection left outside every
 * range: the normative detector cannot see them.  0 on every input the tests hold (asserted there). */
uint64_t orc_rx_presel_miss(const orc_rx_result *r);
void orc_rx_free(orc_rx_result *r);

/* spectrum sensor (predictive_sense.py:72-123,150-268); outputs may be NULL */
int orc_sense_count(const ofdm_sense_cfg *sc, uint64_t nsamples, uint64_t *nmsgs, uint64_t *ndecisions);
int orc_sense(const ofdm_sense_cfg *sc, const ofdm_c32 *iq, uint64_t nsamples, float *msgs, double *mean_inorder,
              uint8_t *bits_inorder, char *hex);
/* sense_loop's tail alone, from ready-made message bodies (pinned by the recorded run logs) */
int orc_sense_decide(const ofdm_sense_cfg *sc, const float *msgs, uint64_t nmsgs, double *mean_inorder,
                     uint8_t *bits_inorder, char *hex);

#ifdef __cplusplus
}
#endif
#endif
