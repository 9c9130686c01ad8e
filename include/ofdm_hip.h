/*
 * ofdm_hip.h -- C ABI of libofdm_hip.so, the MI355X (gfx950) OFDM TX/RX engine.
 *
 * This is the drop-in boundary for the ofdm_mod / ofdm_demod hot path of
 * rubiruchi/ofdm_uhd.  The reference reaches its DSP through SWIG proxies of
 * GNU Radio 3.6 blocks (digital_swig.py); each entry point below names the
 * reference interface it replaces.  Plain C types only: pointers, sizes and
 * one POD configuration struct -- no torch / C++ types.
 *
 * Conventions
 *   - every function returns 0 (OFDM_OK) or a negative OFDM_E_* code; the
 *     message is available from ofdm_last_error().
 *   - bulk data pointers (payload bytes, IQ samples) are DEVICE pointers when
 *     the handle was created with OFDM_F_DEVICE_PTRS, host pointers otherwise.
 *     Small metadata arrays (offsets, lengths, flags, counters) are always
 *     HOST pointers.
 *   - a handle is single-owner (not thread-safe) and bound to one GPU and one
 *     HIP stream.  Calls return after the stream has drained unless stated.
 *   - IQ samples are interleaved float32 (I,Q) = gr_complex, the format of
 *     gr.file_sink(gr.sizeof_gr_complex, ...) (ofdm.py:124-131).
 */
#ifndef OFDM_HIP_H
#define OFDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDM_ABI_VERSION 5

#define OFDM_MAX_FFT 4096
#define OFDM_MAX_CARRIER_HEX 1024 /* hex digits of a carrier map: OFDM_MAX_FFT / 4 */
#define OFDM_MAX_TAPS 512
#define OFDM_MAX_ARITY 256
#define OFDM_MASK_LEN 4096      /* len(random_mask_tuple), ofdm_packet_utils.py:195 */
#define OFDM_MAX_PKT_LEN 4096   /* MAX_PKT_LEN of digital_ofdm_frame_sink */

enum {
  OFDM_OK = 0,
  OFDM_E_INVAL = -1,     /* bad argument / configuration (std::invalid_argument in GR ctors) */
  OFDM_E_NOMEM = -2,
  OFDM_E_CAPACITY = -3,  /* caller-provided output buffer too small */
  OFDM_E_HIP = -4,       /* HIP runtime error */
  OFDM_E_OVERFLOW = -5   /* degenerate input exceeded an internal bound */
};

enum {
  OFDM_F_DEVICE_PTRS = 1u << 0, /* bulk pointers are device pointers */
  OFDM_F_PAD_FOR_USRP = 1u << 1 /* make_packet(pad_for_usrp=True), ofdm.py:45,144 */
};

typedef struct ofdm_c32 {
  float re, im;
} ofdm_c32;

/*
 * Everything ofdm_mod.__init__ / ofdm_demod.__init__ / ofdm_receiver.__init__
 * compute before building their flow graphs (ofdm.py:63-101,204-247,
 * ofdm_receiver.py~:69-98).  The Python host fills it; the engine copies it.
 */
typedef struct ofdm_cfg {
  uint32_t struct_size; /* sizeof(ofdm_cfg), ABI guard */
  int32_t device_id;    /* HIP device ordinal */
  uint32_t flags;       /* OFDM_F_* */

  uint32_t fft_length;     /* options.fft_length      (ofdm.py:64)  power of two, 64..4096 */
  uint32_t occupied_tones; /* options.occupied_tones  (ofdm.py:65)  */
  uint32_t cp_length;      /* options.cp_length       (ofdm.py:66)  */
  uint32_t arity;          /* len(rotated_const)      (ofdm.py:91-92) */

  ofdm_c32 constellation[OFDM_MAX_ARITY]; /* rotated_const (ofdm.py:98-101) */
  ofdm_c32 known_symbol[OFDM_MAX_FFT];    /* ksfreq, occupied_tones entries (ofdm.py:73-77) */

  float tx_amplitude; /* transmit_path._tx_amplitude after the [0,1] clamp (transmit_path.py:56-62) */

  float phase_gain;           /* ofdm_frame_sink phase_gain 0.25 (ofdm.py:238) */
  float freq_gain;            /* ofdm_frame_sink freq_gain  0.25*0.25/4 (ofdm.py:239) */
  float eq_gain;              /* digital_ofdm_frame_sink d_eq_gain 0.05 */
  uint32_t max_fft_shift_len; /* ofdm_frame_acquisition max_fft_shift_len 4 (digital_swig.py:4318-4328) */
  uint32_t sampler_timeout;   /* ofdm_sampler timeout 1000 (digital_swig.py:4719-4724) */

  float peak_rise;  /* gr_peak_detector_fb threshold_factor_rise 0.20 (ofdm_sync_pn) */
  float peak_fall;  /* gr_peak_detector_fb threshold_factor_fall 0.20 */
  float peak_alpha; /* gr_peak_detector_fb alpha 0.001; accepted range (0, 0.25] */

  uint32_t ntaps;             /* len(chan_coeffs), odd (ofdm_receiver.py~:71-75) */
  float taps[OFDM_MAX_TAPS];  /* gr.firdes.low_pass(1, 1, bw+tb, tb, WIN_HAMMING) as float32 */

  uint8_t whitening_mask[OFDM_MASK_LEN]; /* random_mask_tuple (ofdm_packet_utils.py:195-452) */
  uint32_t whitener_offset;              /* make_packet whitener_offset, 0..15 (ofdm_packet_utils.py:100) */

  uint64_t pad_seed; /* seed of the counter-based generator that replaces the mapper's rand()%arity fill */
  /* data-carrier map as a hex string, NUL terminated; "" = the mapper's built-in "FE7F"
   * (transmit_path.py:64).  Used identically by the mapper (container fft_length) and the
   * frame sink (container occupied_tones); see ofdm_set_carrier_map. */
  char carrier_map[OFDM_MAX_CARRIER_HEX + 8];

  /* ofdm_receiver's SYNC selector (ofdm_receiver.py~:89-119).  OFDM_SYNC_PN is the reference's hard-wired choice;
   * OFDM_SYNC_FIXED is its "for testing only" branch (:108-119): chan_filt = multiply_const(1.0), a timing flag on
   * the last sample of every fixed_nsymbols-th symbol starting with the first (ofdm_sync_fixed), a constant
   * frequency-offset input to the NCO.  ("ml" and "pnac" need blocks the reference's tree does not hold.) */
  uint32_t sync_mode;
  uint32_t fixed_nsymbols;   /* symbols per packet incl. the preamble (reference: 18) */
  float fixed_freq_offset;   /* reference: 0.0 */
  uint32_t reserved0;
} ofdm_cfg;
enum { OFDM_SYNC_PN = 0, OFDM_SYNC_FIXED = 1 };

/* Synthetic channel fused into the TX store (replaces the UHD sink/source pair
 * usrp_transmit_path.py:66-72 / usrp_receive_path.py:67-73 for loopback). */
typedef struct ofdm_chan {
  float sigma;             /* AWGN: y = x + sigma*(g1 + j g2)/sqrt(2), g ~ N(0,1) */
  float cfo;               /* carrier offset, radians per sample: x[n] *= exp(j*cfo*n) */
  uint64_t seed;           /* Philox-2x32-7: counter = sample index / 2, key = hash(seed, stream_id) */
  uint64_t stream_id;      /* independent noise per stream */
  uint64_t lead_samples;   /* noise-only samples before the first packet */
  uint64_t tail_samples;   /* noise-only samples after the last packet  */
} ofdm_chan;

/* per-call counters; the cross-GPU reduce sums these */
typedef struct ofdm_stats {
  uint64_t symbols;          /* OFDM symbols processed (TX: emitted incl. preambles; RX: demodulated + preambles) */
  uint64_t samples;          /* IQ samples produced / consumed */
  uint64_t peaks;            /* timing flags raised by the peak detector */
  uint64_t frames;           /* preambles accepted by the sampler */
  uint64_t headers_ok;       /* frames whose 2x16-bit header halves matched */
  uint64_t packets;          /* messages the frame sink posted */
  uint64_t crc_ok;           /* packets whose CRC-32 checked */
  uint64_t chained_frames;   /* frames consumed as payload of an earlier unfinished packet */
  uint64_t overflow;         /* non-zero: an internal bound was hit, result incomplete */
} ofdm_stats;

typedef struct ofdm_handle ofdm_handle;

/* --- lifecycle ---------------------------------------------------------- */
int ofdm_abi_version(void);
int ofdm_device_count(void);
/* replaces ofdm_mod.__init__/ofdm_demod.__init__ graph construction (ofdm.py:45-131,186-261) */
int ofdm_create(const ofdm_cfg *cfg, ofdm_handle **out);
void ofdm_destroy(ofdm_handle *h);
const char *ofdm_last_error(const ofdm_handle *h); /* h may be NULL: error of the last failed ofdm_create */
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the handle's own stream */
int ofdm_set_stream(ofdm_handle *h, void *hip_stream);
/* transmit_path.set_tx_amplitude (transmit_path.py:56-62); clamps to [0,1] */
int ofdm_set_tx_amplitude(ofdm_handle *h, float ampl);
/* digital_ofdm_mapper_bcv::reset_carrier_map of the reference's patched GNU Radio
 * (transmit_path.py:64-70; the call is commented out at :67, so the stock behaviour is
 * never to change it): rebuilds the mapper's and the frame sink's subcarrier tables from a
 * hex string such as the one hex_conv returns (clipped to occupied_tones/4 digits,
 * sensing_and_tramsmitting.py:470).  NULL or "" restores "FE7F".  OFDM_E_INVAL when the
 * string holds a non-hex digit or allocates more carriers than occupied_tones (the
 * blocks' std::invalid_argument); the previous map then stays in force. */
int ofdm_set_carrier_map(ofdm_handle *h, const char *hex);
/* channel applied inside ofdm_tx; NULL disables it */
int ofdm_set_channel(ofdm_handle *h, const ofdm_chan *chan);

/* --- packet framing ------------------------------------------------------
 * Batched make_packet / unmake_packet (ofdm_packet_utils.py:99-143,169-191)
 * incl. digital_crc32 (digital_swig.py:3151-3169).  Framed packet k occupies
 * framed[framed_off[k] .. +payload_len[k]+9 (+USRP pad)].                  */
int ofdm_framed_len(const ofdm_handle *h, uint32_t payload_len, uint32_t *framed_len);
int ofdm_make_packets(ofdm_handle *h, const uint8_t *payloads, const uint64_t *payload_off,
                      const uint32_t *payload_len, int npkt, uint8_t *framed, uint64_t framed_cap,
                      uint64_t *framed_off /* npkt+1, host, out */);

/* --- transmit: ofdm_mod.send_pkt ... multiply_const (ofdm.py:106-118,133-148;
 *     transmit_path.py:47-54) ------------------------------------------------
 * payload bytes in -> make_packet -> ofdm_mapper_bcv -> ofdm_insert_preamble
 * -> fft_vcc(inverse, shift) -> ofdm_cyclic_prefixer -> 1/sqrt(N) -> amp
 * [-> channel].  Packet k starts at sample lead + sym_off[k]*(N+CP).         */
int ofdm_tx_frame_count(const ofdm_handle *h, const uint32_t *payload_len, int npkt,
                        uint64_t *nsymbols, uint64_t *nsamples /* incl. channel lead/tail */);
int ofdm_tx(ofdm_handle *h, const uint8_t *payloads, const uint64_t *payload_off,
            const uint32_t *payload_len, int npkt, ofdm_c32 *iq_out, uint64_t iq_cap,
            uint64_t *nsamples, ofdm_stats *stats /* may be NULL */);

/* the same, returning as soon as the work is queued on the handle's stream: the samples are complete after
 * ofdm_wait() -- or for whatever the caller queues behind them on that stream, e.g. ofdm_rx() on the same
 * handle reading iq_out (device-pointer mode: a TX -> RX loopback then has no host round trip in between).
 * The host metadata arrays may be reused at once; in host-pointer mode iq_out must not be read before ofdm_wait.
 * One asynchronous call may be outstanding per handle: ofdm_wait (or any synchronous call on the handle, all
 * of which end with a stream synchronisation) must come before the next ofdm_tx_async. */
int ofdm_tx_async(ofdm_handle *h, const uint8_t *payloads, const uint64_t *payload_off,
                  const uint32_t *payload_len, int npkt, ofdm_c32 *iq_out, uint64_t iq_cap,
                  uint64_t *nsamples, ofdm_stats *stats /* may be NULL */);
int ofdm_wait(ofdm_handle *h);

/* standalone channel on an existing IQ buffer (same generator as the fused one;
 * sample n of the buffer is stream sample index0+n) */
int ofdm_channel(ofdm_handle *h, ofdm_c32 *iq, uint64_t n, const ofdm_chan *chan, uint64_t index0);

/* --- receive: ofdm_demod (ofdm.py:221-261) = ofdm_receiver (ofdm_receiver.py~:131-142)
 *     + ofdm_frame_sink + _queue_watcher_thread/unmake_packet (ofdm.py:300-305) ---
 * One contiguous IQ stream in; for every message the frame sink would post,
 * in stream order: payload bytes (CRC stripped), length, CRC verdict.  These
 * are exactly the (ok, payload) pairs the reference hands to its callback.  */
int ofdm_rx(ofdm_handle *h, const ofdm_c32 *iq, uint64_t nsamples, uint8_t *payload_out,
            uint64_t payload_cap, uint64_t *payload_off /* max_pkts+1, host */,
            uint32_t *payload_len /* max_pkts, host */, uint8_t *crc_ok /* max_pkts, host */,
            int max_pkts, int *npkt, ofdm_stats *stats /* may be NULL */);

/* digital_ofdm_frame_acquisition::snr() (digital_swig.py:4231-4239, "Return an estimate of the SNR of the channel").
 * In GNU Radio 3.6.0 the block initialises d_snr_est to 0 in its constructor and no code path ever updates it
 * [GR-3.6.0, recalled; the reference never calls it]: the accessor returns that constant, and so does this. */
int ofdm_rx_snr(const ofdm_handle *h, float *snr_est);

/* --- spectrum sensing: the `sensor` flowgraph + sense_loop + hex_conv
 *     (predictive_sense.py:72-123,150-268; same code in sensing_and_tramsmitting*.py) ---
 * stream_to_vector(fft_size) -> fft_vcc(fft_size, True, window) -> complex_to_mag_squared
 * -> bin_statistics_f(fft_size, msgq, tune, tune_delay, dwell_delay): after every retune
 * tune_delay vectors are discarded, then the per-bin MAX over dwell_delay vectors is
 * posted as one message (float32[fft_size], FFT order).  sense_loop then sums avg_msgs
 * messages in float64 (:168-172), consumes skip_msgs more without using them (the message
 * that reaches the else branch, :174), divides by avg_msgs (:175-176), thresholds
 * (bit = 0 if mean > threshold else 1, :179), swaps the halves into ascending-frequency
 * order (:193-205) and packs nibbles LSB-first into upper-case hex (hex_conv :235-268). */
#define OFDM_SENSE_MAX_FFT 4096
typedef struct ofdm_sense_cfg {
  uint32_t struct_size;    /* = sizeof(ofdm_sense_cfg) */
  uint32_t fft_size;       /* power of two, 64..4096 (-s/--fft-size, default 256) */
  uint32_t tune_delay;     /* vectors dropped per message period (>= 0)           */
  uint32_t dwell_delay;    /* vectors max-held per message (>= 1)                 */
  uint32_t avg_msgs;       /* messages averaged per decision (10)                 */
  uint32_t skip_msgs;      /* messages consumed unused per decision (1)           */
  double threshold;        /* 1e-4 (:179); 1e-3 / 0.2 in the secondary_tx variants */
  float window[OFDM_SENSE_MAX_FFT]; /* fft_vcc window taps (window.blackmanharris) */
} ofdm_sense_cfg;

/* how many messages / decisions a stream of nsamples yields */
int ofdm_sense_count(const ofdm_sense_cfg *sc, uint64_t nsamples, uint64_t *nmsgs, uint64_t *ndecisions);
/* iq follows OFDM_F_DEVICE_PTRS; every output is HOST memory and may be NULL:
 * msgs[nmsgs][fft_size] (bin_statistics_f message bodies, FFT order), then per decision
 * mean_inorder[fft_size] (float64, ascending frequency), bits_inorder[fft_size] (0/1),
 * hex[fft_size/4] (no terminator). */
int ofdm_sense(ofdm_handle *h, const ofdm_sense_cfg *sc, const ofdm_c32 *iq, uint64_t nsamples,
               float *msgs, uint64_t msgs_cap /* messages */, double *mean_inorder, uint8_t *bits_inorder,
               char *hex, uint64_t dec_cap /* decisions */, uint64_t *nmsgs, uint64_t *ndecisions);
/* sense_loop alone (:150-222) over ready-made message bodies msgs[nmsgs][fft_size]
 * (HOST memory, e.g. drained from a real gr.msg_queue): same three decision outputs. */
int ofdm_sense_decide(ofdm_handle *h, const ofdm_sense_cfg *sc, const float *msgs, uint64_t nmsgs,
                      double *mean_inorder, uint8_t *bits_inorder, char *hex, uint64_t dec_cap,
                      uint64_t *ndecisions);
/* fuse the sensor into ofdm_rx (BASELINE config 5): while set, every ofdm_rx call also
 * runs the sensing kernels over the same IQ buffer on a second stream, overlapped with
 * the receiver; fetch the outcome of the last call with ofdm_rx_sense_result (same
 * outputs as ofdm_sense).  sc == NULL switches it off. */
int ofdm_set_rx_sense(ofdm_handle *h, const ofdm_sense_cfg *sc);
int ofdm_rx_sense_result(ofdm_handle *h, float *msgs, uint64_t msgs_cap, double *mean_inorder,
                         uint8_t *bits_inorder, char *hex, uint64_t dec_cap, uint64_t *nmsgs,
                         uint64_t *ndecisions);
/* cooperative sensing across GPUs (BASELINE config 5): the DEVICE address of the message
 * bodies of the last run, float32[nmsgs][fft_size], so that the caller can max-reduce them
 * in place over RCCL (ncclMax; powers are >= 0) ... */
int ofdm_sense_device_msgs(ofdm_handle *h, void **d_msgs, uint64_t *nmsgs, uint32_t *fft_size);
/* ... and re-take the decisions (mean / bits / hex) from the reduced bodies; read them with
 * ofdm_rx_sense_result.  Both calls order themselves after the sensing kernels. */
int ofdm_sense_redecide(ofdm_handle *h, const ofdm_sense_cfg *sc);

/* --- chunked streams -------------------------------------------------------------
 * ofdm_rx treats each call as one stream that starts at its first sample (filter and
 * correlator history zero, detector average 0, NCO phase 0), as the reference's flow graph
 * does at start-up.  A continuous capture is fed in overlapping chunks; these three entry
 * points give the caller what it needs to stitch them so that the result equals one call on
 * the whole capture (ofdm_uhd_amd/ofdm.py: ofdm_demod.feed / flush do exactly that):
 *  - the flag sample (last sample of the preamble symbol, relative to the call's iq) of every
 *    packet the last call delivered, in delivery order;
 *  - the flags of the last call with the NCO phase and per-sample phase step in force from
 *    each flag on: phi[n] = phase_j + step_j * (n - flag_j + 1)  (gr_frequency_modulator_fc
 *    driven by the sample-and-held sync angle, ofdm_receiver.py~:97-124).  phase_j is an
 *    integer, units of 2^-64 turn: phases add modulo one turn without rounding, which is what
 *    makes chunked and one-shot processing agree to the last bit; swallowed_j != 0 says the
 *    flag's frame was consumed as payload of a packet that began at an earlier flag;
 *  - the settled past for the following calls: the flags at or before trust_after (relative to
 *    the next call's iq, ascending, inside it) that earlier calls found, with their phase
 *    steps and swallowed marks -- they replace whatever that call detects up to trust_after, in the start of its
 *    overlap where its own detector has not settled -- and the NCO line (phase, step at
 *    pred_flag, which may be negative) of the flag before them, in force up to the call's
 *    first flag.  enable = 0 returns to independent calls.
 *  - ofdm_rx_set_origin: index, in the whole capture, of the first sample the following ofdm_rx calls are
 *    given (default 0).  gr_fft_filter_ccc (ofdm_receiver.py~:76) works in blocks that start at multiples of its
 *    block length counted from the first sample the flow graph ever saw; the engine lays its filter blocks on
 *    that same grid, so a capture handed over in pieces is filtered exactly as one call would filter it. */
int ofdm_rx_set_origin(ofdm_handle *h, uint64_t first_sample_index);
/* Pipelining across batches (one handle = a transmit stream and a receive stream on the GPU): ofdm_rx_submit queues
 * the receiver's input stage (the wait for the transmit batch that fills iq, the channel filter) and returns at
 * once; an ofdm_tx_async issued next is queued behind that stage only -- it may refill the same iq buffer -- and runs
 * while the following ofdm_rx(h, iq, nsamples, ...) (same arguments: it picks the submitted stage up) is busy with
 * its own kernels and host round trips.  Optional: ofdm_rx alone does the same work in order.
 * Exception: with OFDM_SYNC_FIXED (chan_filt is the input itself) or a fused sensor (ofdm_set_rx_sense) the receiver
 * reads iq until the end of ofdm_rx; between ofdm_rx_submit and that ofdm_rx a transmit call whose iq_out overlaps
 * the submitted buffer is refused with OFDM_E_INVAL (transmit into another buffer, or after ofdm_rx). */
int ofdm_rx_submit(ofdm_handle *h, const ofdm_c32 *iq, uint64_t nsamples);
int ofdm_rx_packet_pos(ofdm_handle *h, uint64_t *pos, int cap, int *n);
int ofdm_rx_nco_state(ofdm_handle *h, uint64_t *flags, uint64_t *phase, double *step, uint8_t *swallowed,
                      int cap, int *n);
int ofdm_rx_set_flag_history(ofdm_handle *h, int enable, int n, const int64_t *flags, const double *steps,
                             const uint8_t *swallowed, int64_t trust_after, int64_t pred_flag, uint64_t pred_phase, double pred_step);

/* --- debug taps: the reference's --log probe points (ofdm.py:123-131,253-254;
 *     ofdm_receiver.py~:144-152).  Enable before the call, read after.  Output
 *     is always copied to HOST memory. ---------------------------------------- */
enum {
  OFDM_TAP_TX_PACKETS = 0,   /* uint8: framed packets, concatenated                               */
  OFDM_TAP_TX_FREQ = 1,      /* c32[nsym][N]: ofdm_preambles.dat (mapper+preamble output)        */
  OFDM_TAP_RX_CHAN_FILT = 2, /* c32[nsamples]: ofdm_receiver-chan_filt_c.dat                      */
  OFDM_TAP_RX_METRIC = 3,    /* f32[nsamples]: peak-detector input (M-bar - 1)                    */
  OFDM_TAP_RX_PEAKS = 4,     /* u64[npeaks]: timing-flag sample indices                           */
  OFDM_TAP_RX_ANGLES = 5,    /* f32[npeaks]: sample-and-held angle(P) at each flag                */
  OFDM_TAP_RX_FRAMES = 6,    /* u64[nframes][2]: (flag index, data symbols emitted) per sampler frame */
  OFDM_TAP_RX_FFT = 7,       /* c32[nsym][N]: ofdm_receiver-fft_out_c.dat                         */
  OFDM_TAP_RX_ACQ = 8,       /* c32[nsym][occ]: ofdm_receiver-frame_acq_c.dat                     */
  OFDM_TAP_RX_SINK = 9,      /* c32[ndemapped][occ]: ofdm_frame_sink_c.dat (derotated carriers)   */
  OFDM_TAP_RX_PACKETS = 10,  /* uint8: frame-sink messages before dewhitening, concatenated      */
  OFDM_TAP_TX_MAPPER = 11,   /* c32[ndata][N]: ofdm_mapper_c.dat (mapper output: data symbols only, ofdm.py:124) */
  OFDM_TAP_TX_IFFT = 12,     /* c32[nsym][N]: ofdm_ifft_c.dat (transform output before the cyclic prefix, ofdm.py:128) */
  OFDM_TAP_RX_SAMPLER = 13,  /* c32[nsym][N]: ofdm_receiver-sampler_c.dat (the sampled, derotated symbols = FFT input) */
  OFDM_TAP_RX_SIGMIX = 14,   /* c32[nsamples]: ofdm_receiver-sigmix_c.dat (chan_filt * nco, whole stream)  */
  OFDM_TAP_RX_NCO = 15,      /* c32[nsamples]: ofdm_receiver-nco_c.dat (frequency_modulator_fc output)     */
  OFDM_TAP_RX_PRESEL = 16,   /* f32[nsamples]: the float32 pre-selection of the timing metric (engine-internal stage, DESIGN.md
                              * section 2: it picks the ranges the normative metric is evaluated on and feeds the peak
                              * detector's running average outside them); no reference probe point */
  OFDM_TAP_RX_DEMAPPED = 17, /* u8[nsym]: 1 where the frame sink demapped the symbol (row of RX_FFT / RX_ACQ / RX_SAMPLER): the rows
                              * OFDM_TAP_RX_SINK holds, in order.  Needs OFDM_TAP_RX_SINK enabled. */
  OFDM_TAP_COUNT = 18
};
/* SIGMIX / NCO evaluate the NCO's closed form sample by sample over the whole stream; inside the symbols the
 * sampler picks, the receiver itself advances the same phasor by a float64 recurrence: RX_SAMPLER is
 * bit for bit what the FFT consumed, RX_SIGMIX may differ from it in the last float32 bit. */
int ofdm_set_taps(ofdm_handle *h, uint32_t tap_mask); /* bit i enables OFDM_TAP_i */
int ofdm_tap(ofdm_handle *h, int tap, void *out_host, uint64_t cap_bytes, uint64_t *nbytes);

/* --- measurement: per-kernel HIP-event timing on the handle's stream -------- */
enum {
  OFDM_K_FRAME = 0, /* make_packet: CRC-32 + header + whitening              */
  OFDM_K_TX = 1,    /* map + preamble + IFFT + CP + scale (+channel)         */
  OFDM_K_CHAN = 2,  /* standalone channel                                    */
  OFDM_K_SYNC = 3,  /* Schmidl-Cox metric, float32 pre-selection (streaming) */
  OFDM_K_PEAK = 4,  /* peak detector / sampler / NCO bookkeeping             */
  OFDM_K_DEMOD = 5, /* derotate + FFT + frame acquisition + frame sink       */
  OFDM_K_DEFRAME = 6, /* dewhiten + CRC check + output compaction            */
  OFDM_K_SENSE = 7, /* windowed FFT + |.|^2 + max-hold (+ decision tail)    */
  OFDM_K_FILTER = 8, /* channel filter (overlap-save transforms, streaming)  */
  OFDM_K_EXACT = 9,  /* fixed-point metric + candidates where the pre-selection fired */
  OFDM_K_FRONT = 10, /* fused front end: channel filter + float32 pre-selection in one pass (replaces FILTER + SYNC) */
  OFDM_K_COUNT = 11
};
int ofdm_prof_enable(ofdm_handle *h, int on);
int ofdm_prof_reset(ofdm_handle *h);
int ofdm_prof_get(ofdm_handle *h, int kernel, double *total_ms, uint64_t *launches);
const char *ofdm_kernel_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* OFDM_HIP_H */
