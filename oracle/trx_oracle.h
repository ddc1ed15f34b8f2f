/*
 * trx_oracle.h -- CPU restatement (plain C) of osmo-trx's receive-side burst DSP.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (osmo_trx_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Arithmetic follows the reference's *generic C* operand order (arch/common/convolve_base.c),
 * i.e. BASELINE.json configs[0]; build with -ffp-contract=off (see oracle/Makefile).
 *
 * Parity pinning:
 *   - orc_convolve_{real,complex}: reference golden vectors tests/Transceiver52M/convolve_test_golden.h
 *     and the reference arch kernels compiled unmodified into oracle/_ref (bit-exact, generic build).
 *   - decimator / resampler taps: reference Resampler.cpp compiled unmodified into oracle/_ref.
 *   - detect+demod: the reference's captured burst utils/va-test/nb_chunk_tsc7.cfile with its
 *     expected bits demodbits_tsc7.s8, plus the table/end-to-end anchor values the survey dumped
 *     from the compiled reference.
 *   - sigProcLib.cpp itself is NOT buildable here (needs libosmocore headers); Channelizer needs
 *     FFTW3 (absent): channelizer parity is "unpinned" beyond its mathematical definition.
 */
#ifndef TRX_ORACLE_H
#define TRX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf;

/* CorrType, sigProcLib.h:30-38 */
enum { ORC_OFF = 0, ORC_TSC = 1, ORC_EXT_RACH = 2, ORC_RACH = 3, ORC_SCH = 4, ORC_EDGE = 5, ORC_IDLE = 6 };
/* SignalError, sigProcLib.h:40-46 */
enum { ORC_SIGERR_NONE = 0, ORC_SIGERR_BOUNDS = 1, ORC_SIGERR_CLIP = 2, ORC_SIGERR_UNSUPPORTED = 3, ORC_SIGERR_INTERNAL = 4 };

#define ORC_SINC_TABLESIZE 1024
#define ORC_DELAYFILTS     64
#define ORC_DELAY_HLEN     20
#define ORC_MAX_SEQ        64

/* CorrelationSequence, sigProcLib.cpp:89-103 */
typedef struct {
	int   n;                    /* sequence->size() */
	orc_cf seq[ORC_MAX_SEQ];    /* conjugated, rotated +-1 sequence */
	orc_cf gain;
	float toa;
} orc_corr_seq;

typedef struct {
	float  sinc_table[ORC_SINC_TABLESIZE + 1];          /* sigProcLib.cpp:52,981-988 */
	orc_cf rot1[157], rrot1[157];                       /* GMSKRotation1 / GMSKReverseRotation1 :206-215 */
	orc_cf rot4[625], rrot4[625];                       /* GMSKRotation4 / GMSKReverseRotation4 :195-204 */
	float  pulse1_c0[4];                                /* generateGSMPulse(1) :519-533 */
	float  pulse4_c0[16];                               /* :501-517 */
	float  pulse4_c1[8];                                /* generateC1Pulse :447-458 */
	float  c0_inv[5];                                   /* generateInvertC0Pulse :414-419 */
	orc_corr_seq midamble[8];                           /* gMidambles */
	orc_corr_seq edge_midamble[8];                      /* gEdgeMidambles */
	orc_corr_seq rach[3];                               /* gRACHSequences */
	orc_corr_seq sch;                                   /* gSCHSequence */
	orc_corr_seq dummy;                                 /* gDummySequence */
	float  delay_filt[ORC_DELAYFILTS][ORC_DELAY_HLEN];  /* delayFilters :1005-1044 (real taps) */
	float  dec_taps[16];                                /* Resampler(1,4) partition 0, Resampler.cpp:47-96 */
} orc_tables;

/* estim_burst_params, sigProcLib.h:113-118 */
typedef struct {
	orc_cf  amp;
	float   toa;
	uint8_t tsc;
	float   ci;
} orc_ebp;

/* sigProcLibSetup, sigProcLib.cpp:2139-2172.  Idempotent; returns 1 on success. */
int orc_setup(void);
const orc_tables *orc_get_tables(void);

/* ---- optional arch seam: every FIR of the call graph through the caller's kernels (the reference's own objects in
 * oracle/_ref/libref_generic.so / libref_sse.so); NULL, NULL = the restated generic-C loops.  Tests and bench.py only. ---- */
typedef int (*orc_conv_fn)(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len);
void orc_set_arch(orc_conv_fn conv_real, orc_conv_fn conv_complex);

/* ---- arch kernels (generic C), arch/common/convolve_base.c, convert_base.c ---- */
int  orc_convolve_real(const float *x, int x_len, const float *h, int h_len,
		       float *y, int y_len, int start, int len);
int  orc_convolve_complex(const float *x, int x_len, const float *h, int h_len,
			  float *y, int y_len, int start, int len);
void orc_convert_short_float(float *out, const short *in, int len);
void orc_convert_float_short(short *out, const float *in, float scale, int len);

/* ---- Resampler.cpp ---- */
typedef struct orc_resampler orc_resampler;
orc_resampler *orc_resampler_new(int p, int q, int filt_len, float bw);
void  orc_resampler_free(orc_resampler *r);
const float *orc_resampler_partition(const orc_resampler *r, int path); /* filt_len real taps (reversed) */
/* rotate(): `in` must have filt_len readable samples of history before in[0] */
int   orc_resampler_rotate(const orc_resampler *r, const orc_cf *in, int in_len, orc_cf *out, int out_len);

/* ---- sigProcLib.cpp public functions ---- */
float orc_energy_detect(const orc_cf *burst, int n, unsigned window);                 /* :1573-1585 */
void  orc_vector_slicer(float *dest, const float *src, size_t len);                   /* :546-556   */
/* delayVector :1046-1098, out[n] (may alias nothing) */
void  orc_delay_vector(const orc_cf *in, int n, float delay, orc_cf *out);
/* detectAnyBurst :1926-1957 -> >0 CorrType, 0, or -SignalError */
int   orc_detect_any_burst(const orc_cf *burst, int n, unsigned tsc, float threshold, int sps,
			   int type, unsigned max_toa, orc_ebp *ebp);
/* demodAnyBurst :2130-2137 -> number of soft values written (156 @4sps GMSK, n @1sps, 444 EDGE), <0 error */
int   orc_demod_any_burst(const orc_cf *burst, int n, int type, int sps, orc_ebp *ebp, float *soft);

/* sigProcLib.h:139-148 detectSCHBurst (sch_detect_type in the reference's order) */
enum { ORC_SCH_DETECT_FULL = 0, ORC_SCH_DETECT_NARROW = 1, ORC_SCH_DETECT_BUFFER = 2 };
int   orc_detect_sch_burst(const orc_cf *burst, int n, float thresh, int sps, int state, orc_ebp *ebp);
/* modulateBurst :970-979; returns number of samples written to out (<= 640) */
int   orc_modulate_burst(const uint8_t *bits, int nbits, int guard, int sps, int empty_pulse, orc_cf *out);
/* modulateEdgeBurst(bits, 4, false) :917-936 -> 625 samples */
int   orc_modulate_edge_burst(const uint8_t *bits, int nbits, orc_cf *out);

/* ---- batched "pullRadioVector" DSP core, Transceiver.cpp:724-803 ---- */
typedef struct {
	uint8_t  type;     /* CorrType expected for the slot */
	uint8_t  tsc;
	uint16_t max_toa;
	uint32_t reserved;
} orc_burst_params;

typedef struct {
	int32_t rc;        /* detectAnyBurst return: CorrType (>0), 0, or -SignalError */
	float   toa;       /* symbols */
	float   amp_re, amp_im;
	float   ci;        /* dB */
	float   energy;    /* energyDetect(burst, 20*sps) */
	float   rssi;      /* 20*log10(full_scale/sqrt(energy)) (no rssi_offset) */
	uint8_t tsc;
	uint8_t clip;      /* maxAmplitude > 30000 */
	uint8_t idle;      /* bi->idle */
	uint8_t nbits_div4;/* nbits/4: 37 (148) or 111 (444) when not idle */
} orc_burst_result;

/* iq: n_bursts x burst_len x (I,Q) int16.  soft: n_bursts x soft_stride floats
 * (sliced 0..1 rx_burst when slice!=0, else raw demodAnyBurst output). */
void orc_pull_batch(const int16_t *iq, size_t n_bursts, int burst_len, int sps,
		    const orc_burst_params *params, float threshold, double full_scale,
		    orc_burst_result *res, float *soft, int soft_stride, int slice);
/* the same with n_paths diversity paths per burst (Transceiver.cpp:723-751): iq is n_bursts x n_paths x burst_len x 2 */
void orc_pull_batch_div(const int16_t *iq, size_t n_bursts, int n_paths, int burst_len, int sps,
			const orc_burst_params *params, float threshold, double full_scale,
			orc_burst_result *res, float *soft, int soft_stride, int slice, uint8_t *path);

/* Viterbi alternative (cfg->use_va): Transceiver.cpp:620-645, :782-784 over grgsm_vitac/ */
int   orc_demod_any_burst_va(const orc_cf *burst, int n, int type, int tsc, int max_toa, float scale, float *soft);
void  orc_va_viterbi(const orc_cf *in, unsigned n, const orc_cf *rhh, unsigned start_state,
		     const unsigned *stop_states, unsigned nstops, float *out);

/* TRXD packing, proto_trxd.c:36-66: returns toa_int (1/256 sym), ci centi-bel, soft uint8 */
int     orc_trxd_toa256(double toa);
int16_t orc_trxd_ci_cb(float ci);
void    orc_trxd_soft_u8(uint8_t *dst, const float *rx_burst, unsigned nbits);
/* proto_trxd.c:68-117: the whole v0 / v1 datagram; returns its length (0 = not sent) */
int     orc_trxd_pack(uint8_t *buf, unsigned version, uint32_t fn, uint8_t tn, double rssi, double toa, int idle,
		      int modulation_8psk, uint8_t tss, uint8_t tsc, float ci, const float *rx_burst, unsigned nbits);

/* ---- Transceiver::pullRadioVector() around the DSP core, Transceiver.cpp:665-815: struct bi initialisation, OFF / muted /
 * IDLE early-outs, the 20-entry noise ring (avgVector, radioVector.cpp:79-108), rssi / noise in dBFS, rate counters ---- */
#define ORC_NOISE_CNT 20                                  /* Transceiver.cpp:55 */
typedef struct {
	float    noises[ORC_NOISE_CNT];                   /* avgVector mNoises(NOISE_CNT): value-initialised */
	size_t   itr;
	float    noise_lev;                               /* mNoiseLev(0.0), Transceiver.cpp:66 */
	int      muted;                                   /* mMuted */
	unsigned rx_empty_burst, rx_clipping, rx_no_burst_detected;   /* struct trx_counters, osmo_signal.h:68-70 */
} orc_rx_state;
/* struct trx_ul_burst_ind, proto_trxd.h:24-37 (plain ints for the bool / enum) */
typedef struct {
	float    rx_burst[444];
	unsigned nbits;
	uint32_t fn;
	uint8_t  tn;
	double   rssi, toa, noise;
	int      idle;
	int      modulation;                              /* 0 MODULATION_GMSK, 1 MODULATION_8PSK */
	uint8_t  tss, tsc;
	float    ci;
} orc_ul_burst_ind;
void orc_rx_state_init(orc_rx_state *st);
/* One call of pullRadioVector() given what its DSP calls returned for the slot: `type` = expectedCorrType(), `pow_avg` =
 * sum of the paths' energyDetect() / chans (the argument of :741's sqrt), rc / ebp = detectAnyBurst()'s, `soft` =
 * demodAnyBurst()'s SoftVector (-1..+1, nsoft = 156 GMSK or 444 8-PSK; unused unless rc > 0).  Returns 0, or -2 (-ENOENT). */
int  orc_pull_radio_vector(orc_rx_state *st, int type, uint32_t fn, uint8_t tn, float pow_avg, int rc, const orc_ebp *ebp,
			   const float *soft, int nsoft, double full_scale, double rssi_offset, orc_ul_burst_ind *bi);

/* ---- Channelizer (Channelizer.cpp / ChannelizerBase.cpp), M-path polyphase + M-point DFT ---- */
typedef struct orc_channelizer orc_channelizer;
orc_channelizer *orc_channelizer_new(int m, int block_len, int h_len);
void  orc_channelizer_free(orc_channelizer *c);
const float *orc_channelizer_subfilter(const orc_channelizer *c, int path); /* h_len real taps (reversed) */
/* rotate(): in = block_len*m complex; out[chan] = block_len complex each (chan-major, contiguous) */
int   orc_channelizer_rotate(orc_channelizer *c, const orc_cf *in, int len, orc_cf *out);

#ifdef __cplusplus
}
#endif
#endif
