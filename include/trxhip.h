/*
 * trxhip.h -- C ABI of the MI355X (gfx950) receive-side burst DSP for osmo-trx.
 *
 * This is the thin extern "C" HIP seam that sits where the reference calls its burst DSP:
 *
 *   Transceiver::pullRadioVector()            Transceiver52M/Transceiver.cpp:665-815
 *     energyDetect()                          Transceiver52M/sigProcLib.cpp:1573-1585   (Transceiver.cpp:725)
 *     detectAnyBurst()                        Transceiver52M/sigProcLib.cpp:1926-1957   (Transceiver.cpp:768)
 *     demodAnyBurst()                         Transceiver52M/sigProcLib.cpp:2130-2137   (Transceiver.cpp:786)
 *     vectorSlicer()                          Transceiver52M/sigProcLib.cpp:546-556     (Transceiver.cpp:803)
 *   convert_short_float()                     Transceiver52M/arch/common/convert.h:9    (radioInterface.cpp:344-348)
 *   convolve_real()/convolve_complex()        Transceiver52M/arch/common/convolve.h:6-14
 *   Channelizer::rotate()                     Transceiver52M/Channelizer.cpp:74-99
 *   Resampler::rotate()                       Transceiver52M/Resampler.cpp:131-150
 *
 * The reference runs these one burst at a time on one CPU thread per ARFCN; here they are batched:
 * one call processes N independent bursts that are resident in device memory (HBM).  The
 * single-burst C++ signatures of sigProcLib.h are kept by the host shim in
 * osmo_trx_amd/host/ (batch of 1 over this ABI); INTEGRATION.md shows the binding.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * TRXHIP_E* code and never throws.  Pointers named d_* are DEVICE pointers (hipMalloc / torch
 * CUDA tensors); h_* are host pointers.  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Calls on different contexts/streams may run concurrently from different
 * host threads (one RxUpper thread per ARFCN in the reference, Transceiver.cpp:308-314).
 */
#ifndef TRXHIP_H
#define TRXHIP_H

#include <math.h>     /* powf() in TRXHIP_FAST_CI_ATOL_DB */
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRXHIP_ABI_VERSION 5

/* error codes */
#define TRXHIP_OK          0
#define TRXHIP_EINVAL     (-22)   /* bad argument */
#define TRXHIP_ENOMEM     (-12)
#define TRXHIP_ENODEV     (-19)   /* no usable gfx950 device / HIP runtime failure at init */
#define TRXHIP_EIO        (-5)    /* HIP launch/runtime error (maps to pullRadioVector's -EIO, Transceiver.cpp:686) */
#define TRXHIP_ENOTSUP    (-95)

/* CorrType, sigProcLib.h:30-38 (same numeric values) */
enum trxhip_corr_type {
	TRXHIP_OFF = 0, TRXHIP_TSC = 1, TRXHIP_EXT_RACH = 2, TRXHIP_RACH = 3,
	TRXHIP_SCH = 4, TRXHIP_EDGE = 5, TRXHIP_IDLE = 6
};
/* SignalError, sigProcLib.h:40-46: detect returns -TRXHIP_SIGERR_* */
enum trxhip_signal_error {
	TRXHIP_SIGERR_NONE = 0, TRXHIP_SIGERR_BOUNDS = 1, TRXHIP_SIGERR_CLIP = 2,
	TRXHIP_SIGERR_UNSUPPORTED = 3, TRXHIP_SIGERR_INTERNAL = 4
};

#define TRXHIP_MAX_TOA        112   /* largest max_toa the kernels accept (reference default: 63 AB / 30 NB) */
#define TRXHIP_MAX_BURST_LEN 1536   /* samples per burst the kernels accept (625 @4 SPS, 156/157 @1 SPS) */
#define TRXHIP_BURST_THRESH   4.0f  /* BURST_THRESH, sigProcLib.h:54 */

/* `flags` of the detect/demod entry points */
#define TRXHIP_FLAG_SLICE        1  /* soft bits through vectorSlicer(): 0..1, 148 per burst (else raw -1..+1) */
#define TRXHIP_FLAG_EXACT_DEMOD  2  /* the bit-exact kernel: demodulate with the reference's two FIR stages in its operand order and
                                     * search the TOA with the reference's sums: rc, TOA, amp, C/I and soft bits bit-identical to the
                                     * generic-C reference.  Default (flag clear) is the fused kernel: delay-o-decimate filter with
                                     * FMA (24 of its 35 taps), soft bits within TRXHIP_FUSED_SOFT_ATOL (below) of the reference's,
                                     * ~4x fewer multiply-adds; and the FAST detector (round 5): rc, TSC and TOA still IDENTICAL to
                                     * the reference's -- every early / late decision of peakDetect()'s bisection is either
                                     * certified by a proven rounding-error margin or re-run in the reference's operand order
                                     * (csrc/trx_device.h, peak_detect_fast) -- while the interpolated peak value is an FMA sum:
                                     * amp within TRXHIP_FAST_AMP_RTOL, C/I within TRXHIP_FAST_CI_ATOL_DB(ci) of the reference's.
                                     * Only the 4-SPS / 625-sample kernel has a fused path; others are always exact. */
#define TRXHIP_FLAG_IDLE_DUMMY   4  /* search IDLE slots for the dummy burst, as detectAnyBurst(IDLE) does (detectDummyBurst,
                                     * sigProcLib.cpp:1863-1877, :1945-1947; rc = IDLE on a hit) instead of skipping them as
                                     * pullRadioVector does (Transceiver.cpp:754-755) */
#define TRXHIP_FLAG_USE_VA       8  /* trxhip_hostpipe_cfg.flags only: see there */
#define TRXHIP_FLAG_FEW_NB_SLOTS 16 /* a HINT from a caller that knows its slot types (expectedCorrType() runs on the host,
                                     * Transceiver.cpp:513-601): more than 1/32 of the batch's slots are not normal-burst slots
                                     * (type TSC, tsc < 8, max_toa <= 32).  Results never depend on it; the call then runs the
                                     * general kernel alone instead of the normal-burst kernel + the general one over what that
                                     * leaves (which reads every burst of another type twice).  The host pipe sets it from the
                                     * parameters it is handed; without a hint the library finds out by itself -- the second
                                     * kernel reports how much it was left, and a context that sees more than 1/32 runs the
                                     * general kernel alone for its next 63 launches before it tries again. */
/* The FAST detector's tolerance statement (fused kernels only; tests quote these).  amp = interpolated correlation peak / gain:
 * one 16-term sum per component, FMA against product-then-sum; |amp - ref| <= TRXHIP_FAST_AMP_RTOL * |ref| (complex distance;
 * proven bound 2.6 * 25 u = 3.9e-6 relative to the correlation's arg-max magnitude, measured <= 3e-7).
 * C/I = 10 log10(C / (S - C)) with C = |peak|^2 / den and S the mean sample power (tree-summed in the FAST detector: within
 * 1.2e-6 of the reference's ordered sum): S - C cancels by the factor 1 + C/I, so relative errors eC of C and eS of S become
 * (eC + eS) (1 + 10^(ci/10)) of the ratio:  |ci - ref| <= 1e-4 dB + 4.35 * (2 * TRXHIP_FAST_AMP_RTOL + 1.2e-6) * (1 + 10^(ci/10)) dB.
 * NaN: S < C (a noise slot whose peak estimate exceeds the mean sample power) makes C / (S - C) negative and C/I NaN in the
 * reference as here (sigProcLib.cpp:1637 has no guard); the tests require the SAME NaN pattern (tests/oracle_lib.py
 * assert_fast_ci).  Where S - C cancels to within the relative errors above -- |S - C| <= 3e-6 S, i.e. C/I beyond +55 dB: a peak that
 * explains the whole sample power to six digits, which no burst with any noise or sample rounding in it reaches -- its sign is not
 * covered by the bar: NaN on one side could meet a large finite value on the other. */
#define TRXHIP_FAST_AMP_RTOL        1e-6f
#define TRXHIP_FAST_CI_ATOL_DB(ci)  (1e-4f + 1.4e-5f * (1.0f + powf(10.0f, (ci) * 0.1f)))
/* The one statement of the default (fused) demodulator's tolerance; tests/, tools/parity_campaign.py, bench.py and DESIGN.md
 * quote these two numbers and nothing else.  ABSOLUTE error of a soft bit against the generic-C reference, on soft bits
 * whose full scale is 1 (raw -1..+1 or sliced 0..1):
 *     |soft - ref| <= TRXHIP_FUSED_SOFT_ATOL * max(1, rms / (4 |amp|))        GMSK, 148 / 156 soft bits
 * rms = sqrt(energyDetect()), amp = the detector's amplitude estimate.  The filter's rounding error is relative to the
 * SAMPLES while the soft bits are scaled by 1/amp: for every real detection rms <= 4 |amp| and the bound is the plain
 * 1e-5; a noise-only slot that passes the detector at C/I < -12 dB has samples many times its amplitude estimate and the
 * bound grows with that ratio (worst seen over 4 x 1M bursts: 1.14e-5 on one value in 155 M).  8-PSK rows (444 soft
 * bits: equaliser gain ~2 behind the filter, three soft bits per symbol) and deliberately extreme inputs (tests' fuzz:
 * saturation, impulses, silence): TRXHIP_FUSED_SOFT_ATOL_8PSK.  The north-star bar is 1e-4. */
#define TRXHIP_FUSED_SOFT_ATOL       1e-5f
#define TRXHIP_FUSED_SOFT_ATOL_8PSK  5e-5f

/* Per-burst input: what pullRadioVector() knows before calling detectAnyBurst()
 * (expectedCorrType() Transceiver.cpp:513-601, mTSC, mMaxExpectedDelayAB/NB :757-758). 8 bytes. */
typedef struct trxhip_burst_params {
	uint8_t  type;      /* enum trxhip_corr_type expected for the slot */
	uint8_t  tsc;       /* training sequence code 0..7 */
	uint16_t max_toa;   /* search window, symbols */
	uint32_t reserved;  /* must be 0 */
} trxhip_burst_params;

/* Per-burst output: estim_burst_params (sigProcLib.h:113-118) + what pullRadioVector() derives. 32 bytes. */
typedef struct trxhip_burst_result {
	int32_t rc;          /* detectAnyBurst(): CorrType (>0) | 0 (nothing found) | -SignalError */
	float   toa;         /* ebp.toa, symbols */
	float   amp_re;      /* ebp.amp */
	float   amp_im;
	float   ci;          /* ebp.ci, dB */
	float   energy;      /* energyDetect(burst, 20*sps) */
	float   rssi;        /* 20*log10(full_scale/sqrt(energy)), dBFS without rssi_offset (Transceiver.cpp:751) */
	uint8_t tsc;         /* ebp.tsc */
	uint8_t clip;        /* maxAmplitude() > 30000 (sigProcLib.cpp:1746-1750) */
	uint8_t idle;        /* bi->idle */
	uint8_t nbits_div4;  /* bi->nbits / 4: 37 (148 GMSK), 111 (444 8-PSK), 0 when idle */
} trxhip_burst_result;

typedef struct trxhip_ctx trxhip_ctx;   /* one per device; owns the device-resident tables */

/* ---- lifetime: sigProcLibSetup()/sigProcLibDestroy(), sigProcLib.h:57-60; convolve_init()/convert_init() ---- */
int  trxhip_abi_version(void);
int  trxhip_device_count(void);                       /* number of visible HIP devices (0 if none) */
/* Generates all tables on the host exactly as sigProcLibSetup() does (sigProcLib.cpp:2139-2172)
 * and uploads them to device `device`.  Fails with TRXHIP_ENODEV when no GPU is usable: there is
 * no CPU fallback. */
int  trxhip_create(trxhip_ctx **out, int device);
void trxhip_destroy(trxhip_ctx *ctx);
const char *trxhip_strerror(int err);
/* Work distribution of the 4-SPS kernel for large batches: 1 (default) = the last eighth of the 16-burst groups is drawn from
 * a device-wide counter by whichever CU gets there (evens out the eight dies), 0 = every group dealt statically.  Results
 * never depend on it (tests/test_gpu_parity.py); a measurement switch.  TRXHIP_NO_POOL in the environment makes 0 the
 * default of contexts created afterwards. */
int  trxhip_set_work_pool(trxhip_ctx *ctx, int enabled);
/* Kernel split of trxhip_detect_demod_batch() for the call pullRadioVector() makes (int16 bursts of 625 samples at 4 SPS, fused
 * demodulator, TRXHIP_FLAG_SLICE alone, soft_stride 148): 1 (default) = the normal-burst kernel (csrc/trx_kernel_nb.hip: TSC
 * slots with max_toa <= 32 and nothing else) runs over the batch and the general kernel over the list of bursts it left --
 * slots of other types, wide windows, the rare bursts outside its straight-line paths; 0 = the general kernel alone, as in
 * rounds 1-5.  Results are bit-identical either way (tests/test_gpu_nb_kernel.py); a measurement switch.  TRXHIP_NO_NB_KERNEL in
 * the environment makes 0 the default of contexts created afterwards. */
int  trxhip_set_nb_kernel(trxhip_ctx *ctx, int enabled);
/* Counters of the fused kernels' FAST detector on the context's device since the last reset (synchronises the device):
 * out4[0] = bursts whose TOA search found an uncertified early / late decision on its path and was re-run in the
 * reference's operand order; out4[1..3] reserved.  Diagnostics (tests and tools report the re-run rate). */
int  trxhip_fast_stats(trxhip_ctx *ctx, uint64_t *out4, int reset);

/* ---- table blob: generated once on rank 0, broadcast to the other ranks (RCCL), adopted there ---- */
size_t trxhip_tables_size(void);                                      /* bytes of the device table blob */
int  trxhip_tables_generate_host(void *h_blob, size_t size);          /* host-only: no GPU needed */
int  trxhip_create_from_tables(trxhip_ctx **out, int device, const void *h_blob, size_t size);
int  trxhip_tables_device_ptr(trxhip_ctx *ctx, void **d_blob);        /* for an in-place RCCL broadcast */
uint64_t trxhip_tables_checksum(const void *h_blob, size_t size);     /* FNV-1a over the blob */

/* ---- the hot path: batched pullRadioVector() DSP core ----
 * For each burst b < n_bursts (independent):
 *   int16 IQ -> fp32 (convert_short_float, no scaling) -> energyDetect -> rssi -> clip flag ->
 *   detectAnyBurst(type,tsc,threshold,sps,max_toa) -> if rc>0: demodAnyBurst -> soft bits.
 *   d_iq     : n_bursts * burst_len * 2 int16 (I,Q interleaved, burst-major), 4-byte aligned
 *   d_params : n_bursts trxhip_burst_params
 *   d_results: n_bursts trxhip_burst_result
 *   d_soft   : n_bursts * soft_stride float32 (may be NULL to skip soft output).
 *              TRXHIP_FLAG_SLICE set: rx_burst[] after vectorSlicer(), 0..1, first nbits valid (148) -- what
 *              pullRadioVector() hands to TRXD; clear: raw demodAnyBurst() SoftVector
 *              (-1..+1; 156 values @4 SPS, burst_len @1 SPS).  A burst detected as EDGE (8-PSK) yields 444 soft
 *              bits (nbits_div4 = 111): give soft_stride >= 444 when EDGE slots are possible, otherwise the row is
 *              truncated to soft_stride.  Unused tail and undetected bursts are zero-filled.
 *   flags    : TRXHIP_FLAG_* bits
 *   sps      : 1 or 4; burst_len: 625 @4 SPS (>= 624), 156/157 @1 SPS
 */
int trxhip_detect_demod_batch(trxhip_ctx *ctx,
			      const int16_t *d_iq, const trxhip_burst_params *d_params,
			      trxhip_burst_result *d_results, float *d_soft,
			      size_t n_bursts, int burst_len, int sps,
			      float threshold, float full_scale,
			      int soft_stride, int flags, void *stream);

/* Diversity-path selection in front of the hot path (Transceiver.cpp:723-741): every burst arrives on n_paths receive
 * paths (radioVector::chans()); pullRadioVector() measures energyDetect(path, 20*sps) on each, demodulates the FIRST path
 * with the highest energy and reports rssi / noise from avg = sqrt(sum of the path energies / n_paths).
 *   d_iq_paths   : n_bursts x n_paths x burst_len x 2 int16 (the paths of one burst back to back), 4-byte aligned
 *   d_iq_sel     : n_bursts x burst_len x 2 int16: the chosen path of every burst -> trxhip_detect_demod_batch()
 *   d_avg_energy : n_bursts floats, sum_i pow_i / n_paths (= avg^2)
 *   d_path       : n_bursts chosen path indices (may be NULL)
 * trxhip_apply_diversity_power() then writes energy = avg^2 and rssi = 20*log10(full_scale / avg) into the result records
 * of the detect/demod launch over d_iq_sel (slots that are OFF keep their zeros).  1 <= n_paths <= 8. */
int trxhip_select_diversity_batch(trxhip_ctx *ctx, const int16_t *d_iq_paths, size_t n_bursts, int n_paths, int burst_len, int sps,
				  int16_t *d_iq_sel, float *d_avg_energy, uint8_t *d_path, void *stream);
int trxhip_apply_diversity_power(trxhip_ctx *ctx, trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
				 const float *d_avg_energy, size_t n_bursts, float full_scale, void *stream);

/* Same, from complex64 device samples (the form sigProcLib's detectAnyBurst()/demodAnyBurst() take). */
int trxhip_detect_demod_batch_cf32(trxhip_ctx *ctx,
				   const float *d_iq_cf32, const trxhip_burst_params *d_params,
				   trxhip_burst_result *d_results, float *d_soft,
				   size_t n_bursts, int burst_len, int sps,
				   float threshold, float full_scale,
				   int soft_stride, int flags, void *stream);

/* demodAnyBurst() on its own (sigProcLib.h:151-152): the caller supplies, per burst, the CorrType
 * (d_params[b].type; EDGE selects the 8-PSK demodulator, 444 soft bits, C/I replaced by the EVM estimate) and the
 * estim_burst_params it got from detection as d_ebp[b] = {toa, amp_re, amp_im, unused} (16-byte aligned).
 * Detection is skipped; d_soft receives the soft bits, d_results echoes the parameters. */
int trxhip_demod_batch_cf32(trxhip_ctx *ctx, const float *d_iq_cf32, const trxhip_burst_params *d_params,
			    const float *d_ebp, trxhip_burst_result *d_results, float *d_soft,
			    size_t n_bursts, int burst_len, int sps, int soft_stride, int flags, void *stream);

/* detectSCHBurst() (sigProcLib.h:139-148, sigProcLib.cpp:1805-1861), the MS-side synchronisation-burst search, for
 * n_bufs independent buffers of buf_len complex64 samples at 4 samples per symbol.  `state` is sch_detect_type in the
 * reference's order; the search covers len = 156 (FULL), 8 (NARROW) or 15000 (BUFFER, 12 frames) symbol positions of
 * the first 4*len samples (the reference decimates by 4 whatever `sps` says, :1841), so buf_len >= 4*len is required
 * (the reference asserts, Vector.h:236-237).  sps other than 1 or 4 is the reference's "return -1": -TRXHIP_EINVAL.
 * d_results[b]: rc = 1 / 0 (detectBurst()'s return), toa (symbols, head or 3+39+64 already subtracted, :1853-1858),
 * amp, ci; toa = amp = 0 on a miss (:1846-1850); the other fields are zero. */
enum { TRXHIP_SCH_DETECT_FULL = 0, TRXHIP_SCH_DETECT_NARROW = 1, TRXHIP_SCH_DETECT_BUFFER = 2 };
int trxhip_detect_sch_batch_cf32(trxhip_ctx *ctx, const float *d_iq_cf32, trxhip_burst_result *d_results,
				 size_t n_bufs, size_t buf_len, int sps, int state, float threshold, void *stream);

/* delayVector() (sigProcLib.h:97, sigProcLib.cpp:1046-1098) for n_vec complex64 vectors of `len` samples, one delay
 * (in samples) per vector in d_delays: 64-phase 20-tap fractional filter when |frac| > 0.01, then the integer shift
 * with zero fill.  d_out must not alias d_in. */
int trxhip_delay_vector_batch_cf32(trxhip_ctx *ctx, const float *d_in_cf32, float *d_out_cf32, const float *d_delays,
				   size_t n_vec, int len, void *stream);

/* scaleVector() (sigProcLib.h:94, sigProcLib.cpp:1188-1213): in-place x[i] *= (scale_re + j scale_im) over `len`
 * complex64 samples. */
int trxhip_scale_vector_cf32(trxhip_ctx *ctx, float *d_x_cf32, size_t len, float scale_re, float scale_im, void *stream);

/* The Viterbi alternative of pullRadioVector (cfg->use_va): scaleVector(burst, scale) + demodAnyBurst_va()
 * (Transceiver.cpp:782-784 with scale = 1/16383, :620-645) over gr-gsm's 4-samples-per-symbol MLSE receiver in
 * Transceiver52M/grgsm_vitac/ (get_norm_chan_imp_resp / get_access_imp_resp, detect_burst_nb / _ab, viterbi_detector).
 * d_iq_cf32: n_bursts x burst_len complex64, the burst as that path sees it (it starts 20 samples before the one the
 * detector looks at, Transceiver.cpp:760-762); d_params[b].type TSC selects the normal-burst branch, anything else
 * the access-burst branch (whose Viterbi start state is max_toa, as in the reference; >= 16 selects none); tsc 0..7.
 * d_soft[b][0..soft_stride): +-127 for the 148 (normal) / 88 (access) demodulated bits, 0 behind them; with
 * TRXHIP_FLAG_SLICE through vectorSlicer() (0 / 1).  d_starts (may be NULL): estimated burst start in samples.
 * Samples outside the burst read as 0.  soft_stride >= 148.
 * d_detected (may be NULL): the result records of a preceding detection launch on the same batch; burst b is then
 * demodulated only if d_detected[b].rc > 0, with that rc as the CorrType (Transceiver.cpp:769-784); the others get
 * zeros and start -1.  This chains detection and the Viterbi demodulator on one stream without a host round trip. */
int trxhip_demod_va_batch_cf32(trxhip_ctx *ctx, const float *d_iq_cf32, const trxhip_burst_params *d_params,
			       const trxhip_burst_result *d_detected, float *d_soft, int32_t *d_starts, size_t n_bursts,
			       int burst_len, float scale, int soft_stride, int flags, void *stream);

/* energyDetect() on its own (sigProcLib.h:105, sigProcLib.cpp:1573-1585): mean |x|^2 of `window` samples at
 * stride 4 from sample 0 of each burst (complex64); d_energy: n_bursts floats. */
int trxhip_energy_detect_batch_cf32(trxhip_ctx *ctx, const float *d_iq_cf32, size_t n_bursts, int burst_len,
				    unsigned window, float *d_energy, void *stream);
/* vectorSlicer() (sigProcLib.h:63): dest = clamp(0.5*(src+1), 0, 1), device arrays */
int trxhip_vector_slicer(trxhip_ctx *ctx, float *d_dest, const float *d_src, size_t len, void *stream);

/* TRXD field quantisation into a fixed 156-byte record (a compact device-side format, NOT the wire format -- see
 * trxhip_pack_trxd_wire_batch below), proto_trxd.c:36-66:
 *   d_pkt: n_bursts * 156 bytes: [0..1] toa_int be16 (1/256 sym), [2] rssi u8 (-dBFS), [3..4] ci cB be16,
 *          [5] tsc, [6] idle, [7] nbits/4, [8..155] 148 soft bits uint8 = round(rx_burst*255) */
int trxhip_pack_trxd_batch(trxhip_ctx *ctx, const trxhip_burst_result *d_results, const float *d_soft_sliced,
			   int soft_stride, uint8_t *d_pkt, size_t n_bursts, float rssi_offset, void *stream);

/* TRXD v0 / v1 uplink burst indications in wire format: byte for byte what trxd_send_burst_ind_v0() / _v1()
 * hand to write() (proto_trxd.c:68-117; struct trxd_hdr_common / _v0_specific / _v1_specific, proto_trxd.h:56-106):
 *   [0]     version << 4 | tn & 7             trxd_fill_common()        proto_trxd.c:28-34
 *   [1..4]  fn, big endian
 *   [5]     rssi = (uint8_t) bi->rssi         trxd_fill_v0_specific()   :36-45   (rssi = result.rssi + rssi_offset)
 *   [6..7]  (int)(toa * 256.0 + 0.5), be16
 *   v0:  [8..8+nbits) soft bits, then two trailing bytes (the reference leaves the first uninitialised and zeroes the
 *        second, :83-87; both are 0 here); idle indications are not sent (length 0, :71-73)
 *   v1:  [8] idle << 7 | modulation << 3 | tsc & 7    trxd_fill_v1_specific() :47-60; modulation = GMSK: tss & 3,
 *        8-PSK: 4 | tss & 1 (TRXD_MODULATION_*, proto_trxd.h:84-88); [9..10] (int16)(ci * 10 + 0.5) be16;
 *        [11..11+nbits) soft bits unless idle (:100-109)
 *   soft bits: (uint8_t) round(rx_burst[i] * 255.0), nbits = 148 (GMSK) or 444 (8-PSK)     :62-66
 * An idle indication carries toa = ci = tsc = 0 and GMSK, as pullRadioVector() leaves them (Transceiver.cpp:694-704,
 * ret_idle :808-814).  A slot that is OFF produces nothing (pullRadioVector() returns -ENOENT, :704-707): length 0.
 * rssi outside 0..255 (or NaN) is undefined behaviour in the reference's double -> uint8_t conversion; here it
 * saturates.
 *   d_results, d_params : the records of the detect/demod launch over the same batch (d_params gives OFF)
 *   d_soft_sliced       : its soft output with TRXHIP_FLAG_SLICE, soft_stride >= 148 (>= 444 for 8-PSK rows)
 *   d_meta              : per burst {fn, tn, version 0|1, tss}
 *   d_pkt               : n_bursts x pkt_stride bytes, pkt_stride a multiple of 4 and >= 160 (>= 456 when 8-PSK rows
 *                         are possible; a datagram that does not fit is truncated to pkt_stride); burst b's datagram
 *                         starts at d_pkt + b * pkt_stride, bytes behind its length are 0
 *   d_pkt_len           : n_bursts datagram lengths (0 = nothing to send) */
typedef struct trxhip_trxd_meta {
	uint32_t fn;        /* TDMA frame number */
	uint8_t  tn;        /* timeslot 0..7 */
	uint8_t  version;   /* TRXD header version negotiated for the channel: 0 or 1 (mVersionTRXD[chan]) */
	uint8_t  tss;       /* training sequence set (the reference always sends 0, Transceiver.cpp:703) */
	uint8_t  reserved;
} trxhip_trxd_meta;
#define TRXHIP_TRXD_V0_HDR   8
#define TRXHIP_TRXD_V1_HDR  11
#define TRXHIP_TRXD_MAX_PKT (TRXHIP_TRXD_V1_HDR + 444)
int trxhip_pack_trxd_wire_batch(trxhip_ctx *ctx, const trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
				const float *d_soft_sliced, int soft_stride, const trxhip_trxd_meta *d_meta,
				uint8_t *d_pkt, int pkt_stride, uint16_t *d_pkt_len, size_t n_bursts, float rssi_offset,
				void *stream);

/* ---- host-fed, stream-pipelined form of the hot path (host buffers in, host buffers out) ----
 * pullRadioVector()'s callers hold their bursts in host memory.  A hostpipe owns `depth` staging slots of pinned
 * host memory, each with its own device buffers and HIP stream; a submitted slot runs
 *     H2D (iq, params, meta) -> trxhip_detect_demod_batch [-> trxhip_pack_trxd_wire_batch] -> D2H (results, soft | pkt)
 * asynchronously, so that slot k+1's upload, slot k's kernels and slot k-1's download overlap.  Nothing is
 * allocated after create.  The producer writes bursts straight into trxhip_hostpipe_slot_buffers().iq (no second
 * copy); trxhip_hostpipe_run() is the convenience form for pageable caller buffers (it copies through the slots). */
typedef struct trxhip_hostpipe trxhip_hostpipe;
typedef struct trxhip_hostpipe_cfg {
	uint32_t max_bursts;   /* capacity of one slot */
	int32_t  depth;        /* number of slots, 2..16 */
	int32_t  burst_len;    /* 625 at 4 SPS */
	int32_t  sps;
	int32_t  soft_stride;  /* floats per burst downloaded (148 / 156 / 444); 0 = no float soft output */
	int32_t  pkt_stride;   /* bytes per burst of TRXD datagrams downloaded (multiple of 4, >= 160); 0 = no TRXD packing */
	int32_t  flags;        /* TRXHIP_FLAG_*; TRXD packing implies TRXHIP_FLAG_SLICE.  TRXHIP_FLAG_USE_VA (host pipe only): the
	                        * cfg->use_va flow of pullRadioVector (Transceiver.cpp:760-787) -- the slot's bursts are what the radio
	                        * read 20 samples early; power / rssi come from them as read, detection runs on the copy shifted by 20
	                        * samples (zeros behind), the soft bits from scaleVector(1 / 16383) + demodAnyBurst_va() on the unshifted
	                        * burst (trxhip_demod_va_batch_cf32 chained behind the detection records): hard 0 / 1 rows of 148 */
	float    threshold;    /* TRXHIP_BURST_THRESH */
	float    full_scale;
	float    rssi_offset;  /* enters the TRXD rssi byte only; result.rssi stays without it */
	int32_t  n_paths;      /* diversity paths per burst (0 or 1: none).  > 1: a slot's iq holds max_bursts x n_paths x burst_len
	                        * samples, every submit runs trxhip_select_diversity_batch() first and
	                        * trxhip_apply_diversity_power() behind the detector (Transceiver.cpp:723-751) */
} trxhip_hostpipe_cfg;
typedef struct trxhip_hostpipe_slot {      /* pinned host memory, valid until trxhip_hostpipe_destroy() */
	int16_t             *iq;       /* in : max_bursts x [n_paths x] burst_len x 2 */
	trxhip_burst_params *params;   /* in : max_bursts */
	trxhip_trxd_meta    *meta;     /* in : max_bursts (NULL without TRXD packing) */
	trxhip_burst_result *results;  /* out: max_bursts */
	float               *soft;     /* out: max_bursts x soft_stride (NULL when soft_stride = 0) */
	uint8_t             *pkt;      /* out: max_bursts x pkt_stride  (NULL when pkt_stride = 0) */
	uint16_t            *pkt_len;  /* out: max_bursts */
} trxhip_hostpipe_slot;
int  trxhip_hostpipe_create(trxhip_ctx *ctx, const trxhip_hostpipe_cfg *cfg, trxhip_hostpipe **out);
void trxhip_hostpipe_destroy(trxhip_hostpipe *p);
int  trxhip_hostpipe_slot_buffers(trxhip_hostpipe *p, int slot, trxhip_hostpipe_slot *out);
/* change the scalar parameters of later submits (detection threshold, rxFullScale, rssi_offset) without touching the
 * staging buffers: callers whose channels differ in full scale share one pipe */
int  trxhip_hostpipe_set_levels(trxhip_hostpipe *p, float threshold, float full_scale, float rssi_offset);
/* enqueue slot's first n_bursts bursts; returns at once.  The slot's buffers must not be touched until wait(). */
int  trxhip_hostpipe_submit(trxhip_hostpipe *p, int slot, size_t n_bursts);
/* block until the slot's job has finished (TRXHIP_OK), or return TRXHIP_EIO if it failed */
int  trxhip_hostpipe_wait(trxhip_hostpipe *p, int slot);
/* 0 = finished (or never submitted), 1 = still running, < 0 error */
int  trxhip_hostpipe_query(trxhip_hostpipe *p, int slot);
/* ---- bursts by reference (round 5): no CPU copy of the samples at all ----
 * The reference cuts every burst out of the radio's receive ring with one CPU copy (radioInterface.cpp:272-291,
 * unRadioifyVector into a new radioVector); writing the burst into slot.iq is that copy.  When the ring itself is
 * registered with the pipe, a slot can instead carry one host POINTER per burst: submit_by_ref() uploads only
 * [params][meta], and a device kernel fetches the n bursts from the ring over the link (burst_len x 4 bytes from
 * each pointer, x n_paths with diversity) into the slot's device buffer; everything behind is trxhip_hostpipe_submit().
 * Contract: every src[i] is 4-byte aligned and lies, with its whole burst, inside one registered range (TRXHIP_EINVAL
 * otherwise, nothing enqueued); the samples stay unchanged until wait() on the slot has returned.
 * register_host: pins [base, base + bytes) and maps it into the device's address space (hipHostRegister; a range some
 * other pipe of the process has registered already is shared -- accepted only if that registration is mapped and covers
 * the whole range, TRXHIP_EINVAL otherwise; the sharing pipe does not own the pin: whoever registered first must outlive
 * every pipe that shares the range, which is how a multi-device gatherer destroys its pipes -- in reverse order of creation);
 * round 6: runs of 16 or more addresses a constant step apart are moved by the copy engine (one copy per run), the rest by
 * the fetch kernel; at most 8 ranges per pipe; unregister_host (or destroy)
 * releases what this pipe pinned -- before the memory is freed, and with no slot that refers to the range in flight.
 * register / unregister are not synchronised against submits of the same pipe on other threads. */
int  trxhip_hostpipe_register_host(trxhip_hostpipe *p, const void *base, size_t bytes);
int  trxhip_hostpipe_unregister_host(trxhip_hostpipe *p, const void *base);
/* the slot's pointer array: max_bursts entries of pinned host memory, valid until trxhip_hostpipe_destroy() */
int  trxhip_hostpipe_slot_sources(trxhip_hostpipe *p, int slot, const int16_t ***out_src);
int  trxhip_hostpipe_submit_by_ref(trxhip_hostpipe *p, int slot, size_t n_bursts);
/* Pageable buffers: n bursts are cut into slot-sized chunks, staged, processed with all slots in flight and copied
 * out.  h_meta / h_soft / h_pkt / h_pkt_len may be NULL (must be NULL when the pipe was created without them). */
int  trxhip_hostpipe_run(trxhip_hostpipe *p, const int16_t *h_iq, const trxhip_burst_params *h_params,
			 const trxhip_trxd_meta *h_meta, trxhip_burst_result *h_results, float *h_soft, uint8_t *h_pkt,
			 uint16_t *h_pkt_len, size_t n_bursts);

/* ---- arch kernels, batched (arch/common/convolve.h:6-14, convert.h:9) ----
 * y[b][i] = sum_k x[b][i + start - (h_len-1) + k] * h[k]  (correlation form, no tap flip), b < n_vec.
 * x: n_vec * x_len complex64; the caller guarantees start >= h_len-1 and start+len <= x_len
 * (the reference's bounds_check(), convolve_base.c:88-105 -> returns TRXHIP_EINVAL otherwise).
 * h: h_len complex64 on device (imag ignored for _real). */
int trxhip_convolve_real_batch(trxhip_ctx *ctx, const float *d_x, int x_len, const float *d_h, int h_len,
			       float *d_y, int y_len, int start, int len, size_t n_vec, void *stream);
int trxhip_convolve_complex_batch(trxhip_ctx *ctx, const float *d_x, int x_len, const float *d_h, int h_len,
				  float *d_y, int y_len, int start, int len, size_t n_vec, void *stream);
int trxhip_convert_short_float(trxhip_ctx *ctx, float *d_out, const int16_t *d_in, size_t len, void *stream);
/* convert_float_short() in its generic-C form (convert.h:6, convert_base.c:20-25): out[i] = (short)(in[i] * scale) */
int trxhip_convert_float_short(trxhip_ctx *ctx, int16_t *d_out, const float *d_in, float scale, size_t len, void *stream);
/* cxvec_fft() (fft.h:6-11, fft.c:55-114): `howmany` M-point DFTs, transform t reading d_in[j*istride + t] and writing
 * d_out[k*ostride + t] (complex64 units), forward (reverse = 0) or backward, unnormalised -- the layout of the
 * reference's fftwf_plan_many_dft() call.  d_out must not alias d_in. */
int trxhip_dft_batch(trxhip_ctx *ctx, const float *d_in, float *d_out, int m, size_t howmany, size_t istride, size_t ostride,
		     int reverse, void *stream);

/* ---- Channelizer::rotate (M-path polyphase analysis bank + M-point DFT), batched over blocks ----
 * d_in : n_blocks * block_len * m wideband samples as int16 IQ (a continuous stream; block j's
 *        filter history is the tail of block j-1, zero for block 0 -- Channelizer.cpp:86-88)
 * d_out: m * (n_blocks * block_len) complex64, channel-major (outputBuffer(chan), Channelizer.cpp:60-66) */
int trxhip_channelize_batch(trxhip_ctx *ctx, const int16_t *d_in, float *d_out,
			    size_t n_blocks, int m, int block_len, int h_len, void *stream);
/* Resampler(p,q,16)::rotate over a continuous stream per channel: in n_in samples -> out n_in*p/q */
int trxhip_resample_batch(trxhip_ctx *ctx, const float *d_in, float *d_out, size_t n_in, int p, int q,
			  size_t n_chan, size_t in_stride, size_t out_stride, void *stream);

/* ---- streaming multi-ARFCN receive front end: RadioInterfaceMulti::pullBuffer(), radioInterfaceMulti.cpp:237-314 ----
 * Channelizer(4, block_len, 16)::rotate followed by Resampler(p, q, 16)::rotate on every filterbank channel, called
 * chunk after chunk: the object carries what the reference carries between calls (Channelizer::hist,
 * Channelizer.cpp:86-88; history[lchan], radioInterfaceMulti.cpp:283-300), so any chunking of a stream gives the
 * result of processing it in one piece.  (p,q) = (65,48) in RadioInterfaceMulti; RadioInterfaceResamp uses the
 * same Resampler with (65,96) and (52,75), radioInterfaceResamp.cpp:36-41: block_len must be a multiple of q. */
typedef struct trxhip_rx_frontend trxhip_rx_frontend;
int  trxhip_rx_frontend_create(trxhip_ctx *ctx, int block_len, int p, int q, trxhip_rx_frontend **out);
void trxhip_rx_frontend_destroy(trxhip_rx_frontend *f);
int  trxhip_rx_frontend_reset(trxhip_rx_frontend *f, void *stream);           /* zero the carried history */
/* Start mid-stream (a stream is sharded in time, one shard per GPU, with overlap at the shard edges): make the
 * carried state what it would be had the stream been processed up to the shard's first block.  d_wide_prev = the
 * n_blocks_prev >= 1 blocks (n_blocks_prev * block_len * 4 wideband int16 IQ samples, 16-byte aligned) that immediately
 * precede the shard; one block is enough, because both filters are FIR -- the channelizer carries 15 time steps
 * (Channelizer::hist, Channelizer.cpp:86-88), the resampler 15 channel samples (history[lchan],
 * radioInterfaceMulti.cpp:283-300).  The shard's output is then bit-identical to the same blocks' output of an unsharded
 * run.  n_blocks_prev = 0 (stream start) is trxhip_rx_frontend_reset(). */
int  trxhip_rx_frontend_seed(trxhip_rx_frontend *f, const int16_t *d_wide_prev, size_t n_blocks_prev, void *stream);
/* d_wide: n_blocks * block_len * 4 wideband int16 IQ samples (16-byte aligned);
 * d_out : 4 channels x (n_blocks*block_len*p/q) complex64, channel c at d_out + 2*c*out_stride floats */
int  trxhip_rx_frontend_pull(trxhip_rx_frontend *f, const int16_t *d_wide, size_t n_blocks, float *d_out,
			     size_t out_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRXHIP_H */
