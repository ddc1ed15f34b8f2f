// trx_capi.cpp -- the extern "C" seam declared in include/trxhip.h.
// Thin: argument checks, table upload, kernel launches.  No CPU fallback anywhere: when no HIP
// device is usable trxhip_create() fails with TRXHIP_ENODEV and nothing else can be called.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/trxhip.h"
#include "trx_tables.h"

// kernel launchers (trx_kernels.hip, trx_aux_kernels.hip)
extern "C" int trx_launch_pull(unsigned *d_pool_ctr, const void *d_iq, int cf32, const trxhip_burst_params *d_params,
			       trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, const float *d_ebp_in,
			       size_t n_bursts, int L, int sps, float thresh, float full_scale,
			       int soft_stride, int slice, int n_cu, hipStream_t stream);
extern "C" int trx_launch_pull4_nb(unsigned *d_pool_ctr, const void *d_iq, const trxhip_burst_params *d_params,
				   trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, size_t n_bursts,
				   float thresh, float full_scale, int n_cu, unsigned *d_redo, unsigned *h_left, hipStream_t stream);
extern "C" int trx_unit_masks_match(const trx_tables *t);       /* trx_kernel4.hip: compiled-in sign masks vs the tables */
extern "C" int trx_launch_pack_trxd(const trxhip_burst_result *d_results, const float *d_soft, int soft_stride,
				    uint8_t *d_pkt, size_t n_bursts, float rssi_offset, hipStream_t stream);
extern "C" int trx_launch_convolve(const float *d_x, int x_len, const float *d_h, int h_len, int h_complex,
				   float *d_y, int y_len, int start, int len, size_t n_vec, hipStream_t stream);
extern "C" int trx_launch_convert_short_float(float *d_out, const int16_t *d_in, size_t len, hipStream_t stream);
extern "C" int trx_launch_convert_float_short(int16_t *d_out, const float *d_in, float scale, size_t len, hipStream_t stream);
extern "C" int trx_launch_dft_strided(const float *d_in, float *d_out, int m, size_t howmany, size_t istride, size_t ostride,
				      int reverse, hipStream_t stream);
extern "C" int trx_launch_frontend_fused(const int16_t *d_wide, float *d_out, size_t n_total, int p, int q, size_t out_stride,
					 const float *parts, const trx_tables *d_tab, void *d_wide_hist_io, const void *d_chan_hist_in,
					 void *d_chan_hist_out, hipStream_t stream);
extern "C" int trx_launch_channelize(const int16_t *d_in, float *d_out, size_t n_total, size_t out_stride,
				     const trx_tables *d_tab, void *d_hist_io, hipStream_t stream);
extern "C" int trx_launch_resample(const float *d_in, float *d_out, size_t n_in, int p, int q, size_t n_chan,
				   size_t in_stride, size_t out_stride, const float *d_parts, void *d_hist_io, hipStream_t stream);

extern "C" int trx_launch_energy_detect(const float *d_x, size_t n_bursts, int burst_len, unsigned window, float *d_out,
					hipStream_t stream);
extern "C" int trx_launch_sch_detect(const float *d_iq, size_t buf_stride, trxhip_burst_result *d_results,
				     const trx_tables *d_tab, size_t n_bufs, int len, int start, int toa_sub, float thresh,
				     int unit_tables, hipStream_t stream);
extern "C" int trx_unit_mask_sch_match(const trx_tables *t);   /* trx_sch.hip: the SCH sequence's compiled-in sign mask vs the tables */
extern "C" int trx_launch_delay_vector(const float *d_in, float *d_out, const float *d_delays, const trx_tables *d_tab,
				       size_t n_vec, int len, hipStream_t stream);
extern "C" int trx_launch_scale_vector(float *d_x, size_t len, float sr, float si, hipStream_t stream);
extern "C" int trx_launch_va_demod(const float *d_iq, const trxhip_burst_params *d_params,
				   const trxhip_burst_result *d_detected, float *d_soft, int32_t *d_starts,
				   size_t n_bursts, int L, float scale, int soft_stride, int flags, hipStream_t stream);
extern "C" int trx_launch_vector_slicer(float *d_dst, const float *d_src, size_t len, hipStream_t stream);

#include "trx_ctx.h"

#define TRXHIP_FLAG_DIAG_MASK 0x7fffff00   /* phase-ablation bits of the -DTRX_DIAG profiling build (tools/) */
#define TRXHIP_IFLAG_NO_UNIT  0x40         /* internal: see trx_device.h */
#define TRXHIP_IFLAG_NO_SYM   0x80
#define TRXHIP_IFLAG_NO_FAST  0x20
extern "C" int trx_fast_stats_read(unsigned long long *out4, int reset);   /* trx_kernel4.hip */

extern "C" {

int trxhip_abi_version(void) { return TRXHIP_ABI_VERSION; }

int trxhip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

const char *trxhip_strerror(int err)
{
	switch (err) {
	case TRXHIP_OK: return "ok";
	case TRXHIP_EINVAL: return "invalid argument";
	case TRXHIP_ENOMEM: return "out of device memory";
	case TRXHIP_ENODEV: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
	case TRXHIP_EIO: return "HIP runtime / kernel launch error";
	case TRXHIP_ENOTSUP: return "not supported";
	default: return "unknown error";
	}
}

size_t trxhip_tables_size(void) { return sizeof(trx_tables); }

int trxhip_tables_generate_host(void *h_blob, size_t size)
{
	if (!h_blob || size != sizeof(trx_tables))
		return TRXHIP_EINVAL;
	return trx_tables_generate(static_cast<trx_tables *>(h_blob)) == 0 ? TRXHIP_OK : TRXHIP_EINVAL;
}

uint64_t trxhip_tables_checksum(const void *h_blob, size_t size)
{
	const unsigned char *p = static_cast<const unsigned char *>(h_blob);
	uint64_t h = 1469598103934665603ull;          /* FNV-1a 64 */
	for (size_t i = 0; i < size; i++) {
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

int trxhip_create_from_tables(trxhip_ctx **out, int device, const void *h_blob, size_t size)
{
	if (!out || !h_blob || size != sizeof(trx_tables))
		return TRXHIP_EINVAL;
	const trx_tables *t = static_cast<const trx_tables *>(h_blob);
	if (t->magic != TRX_TABLES_MAGIC || t->version != TRX_TABLES_VERSION)
		return TRXHIP_EINVAL;
	/* the composite-tap window is a build-time choice of generator AND kernels (TRX_FUSED_U0 / TRX_FUSED_NT): a blob made by a
	 * build with another window would be read with shifted taps */
	if (t->fused_u0 != TRX_FUSED_U0 || t->fused_nt != TRX_FUSED_NT)
		return TRXHIP_EINVAL;
	/* the exact delay filters of the kernels skip taps 0, 17, 18, 19: exactly 0.0f in every filter sigProcLibSetup() can
	 * generate (the sinc LUT is zero beyond 8 pi, sigProcLib.cpp:990-998, :2164-2172).  A blob that breaks this is not such a
	 * table set: refused rather than silently evaluated with 16 of its 20 taps.  (Non-finite samples: the
	 * reference's x * 0 turns an Inf / NaN inside the 20-tap span into NaN where the skipped taps drop it -- complex64 input
	 * only; int16 input is always finite.) */
	for (int f = 0; f < TRX_DELAY_FILTS; f++)
		if (t->delay_filt[f][0] != 0.0f || t->delay_filt[f][17] != 0.0f || t->delay_filt[f][18] != 0.0f || t->delay_filt[f][19] != 0.0f)
			return TRXHIP_ENOTSUP;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
		return TRXHIP_ENODEV;
	if (hipSetDevice(device) != hipSuccess)
		return TRXHIP_ENODEV;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return TRXHIP_ENODEV;

	trxhip_ctx *ctx = new (std::nothrow) trxhip_ctx;
	if (!ctx)
		return TRXHIP_ENOMEM;
	ctx->device = device;
	ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	ctx->d_tables = nullptr;
	ctx->no_unit = trx_unit_masks_match(t) ? 0 : 1;
	ctx->sch_unit = trx_unit_mask_sch_match(t) ? 1 : 0;
	ctx->d_pool = nullptr;
	ctx->pool_next.store(0u);
	ctx->pool_enabled = getenv("TRXHIP_NO_POOL") ? 0 : 1;          /* measurement switch, read once per context */
	ctx->nb_enabled = getenv("TRXHIP_NO_NB_KERNEL") ? 0 : 1;       /* the same for the normal-burst kernel (A/B against the general one) */
	ctx->redo_next = 0;
	for (int i = 0; i < TRX_REDO_SLOTS; i++) {
		ctx->redo[i].d = nullptr;
		ctx->redo[i].cap = 0;
		ctx->redo[i].ev = nullptr;
		ctx->redo[i].busy = 0;
		ctx->redo[i].h_left = nullptr;
		ctx->redo[i].n_last = 0;
	}
	ctx->split_backoff = 0;
	ctx->no_backoff = getenv("TRXHIP_NO_BACKOFF") ? 1 : 0;         /* measurement switch: the split path whatever it leaves */
	ctx->no_sym = 0;                                           /* the straight-line decimator reads taps 0..7 and mirrors them */
	for (int k = 0; k < 8; k++)
		if (memcmp(&t->dec_taps[k], &t->dec_taps[15 - k], sizeof(float)) != 0)
			ctx->no_sym = 1;
	/* the FAST detector's margin assumes sum_u |w_u| <= 2.6 for the 16 sinc weights of every fractional position
	 * (trx_device.h, TRX_FAST_W): taps i <= fl use q = 512 k + f, taps i > fl use q = 512 k + (512 - f), k = 0 .. 7 */
	ctx->no_fast = 0;
	for (int f = 0; f < 512 && !ctx->no_fast; f++) {
		double sum = 0.0;
		for (int k = 0; k < 8; k++) {
			const int qa = 512 * k + f, qb = 512 * k + (512 - f);
			sum += fabs((double)t->sincv[trx_sincv_swz(qa)]);
			if (qb < TRX_SINCV_LEN)
				sum += fabs((double)t->sincv[trx_sincv_swz(qb)]);
		}
		if (!(sum <= 2.6))
			ctx->no_fast = 1;
	}
	if (hipMalloc(reinterpret_cast<void **>(&ctx->d_tables), sizeof(trx_tables)) != hipSuccess) {
		delete ctx;
		return TRXHIP_ENOMEM;
	}
	if (hipMemcpy(ctx->d_tables, h_blob, sizeof(trx_tables), hipMemcpyHostToDevice) != hipSuccess) {
		(void)hipFree(ctx->d_tables);
		delete ctx;
		return TRXHIP_EIO;
	}
	if (hipMalloc(reinterpret_cast<void **>(&ctx->d_pool), TRX_POOL_SLOTS * 64) != hipSuccess ||
	    hipMemset(ctx->d_pool, 0, TRX_POOL_SLOTS * 64) != hipSuccess) {
		if (ctx->d_pool) (void)hipFree(ctx->d_pool);
		ctx->d_pool = nullptr;                                 /* (the kernels run without the pool) */
	}
	*out = ctx;
	return TRXHIP_OK;
}

int trxhip_create(trxhip_ctx **out, int device)
{
	if (!out)
		return TRXHIP_EINVAL;
	trx_tables *t = static_cast<trx_tables *>(malloc(sizeof(trx_tables)));
	if (!t)
		return TRXHIP_ENOMEM;
	int rc = trx_tables_generate(t) == 0 ? TRXHIP_OK : TRXHIP_EINVAL;
	if (rc == TRXHIP_OK)
		rc = trxhip_create_from_tables(out, device, t, sizeof(trx_tables));
	free(t);
	return rc;
}

void trxhip_destroy(trxhip_ctx *ctx)
{
	if (!ctx)
		return;
	if (hipSetDevice(ctx->device) == hipSuccess) {
		if (ctx->d_tables) (void)hipFree(ctx->d_tables);
		if (ctx->d_pool) (void)hipFree(ctx->d_pool);
		for (int i = 0; i < TRX_REDO_SLOTS; i++) {
			if (ctx->redo[i].busy) (void)hipEventSynchronize(ctx->redo[i].ev);
			if (ctx->redo[i].ev) (void)hipEventDestroy(ctx->redo[i].ev);
			if (ctx->redo[i].d) (void)hipFree(ctx->redo[i].d);
			if (ctx->redo[i].h_left) (void)hipHostFree(ctx->redo[i].h_left);
		}
	}
	delete ctx;
}

int trxhip_set_nb_kernel(trxhip_ctx *ctx, int enabled)
{
	if (!ctx)
		return TRXHIP_EINVAL;
	ctx->nb_enabled = enabled ? 1 : 0;
	return TRXHIP_OK;
}

int trxhip_set_work_pool(trxhip_ctx *ctx, int enabled)
{
	if (!ctx)
		return TRXHIP_EINVAL;
	ctx->pool_enabled = enabled ? 1 : 0;
	return TRXHIP_OK;
}

int trxhip_fast_stats(trxhip_ctx *ctx, uint64_t *out4, int reset)
{
	if (!ctx || !out4)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	unsigned long long v[4] = { 0, 0, 0, 0 };
	if (trx_fast_stats_read(v, reset) != 0)
		return TRXHIP_EIO;
	for (int k = 0; k < 4; k++)
		out4[k] = v[k];
	return TRXHIP_OK;
}

int trxhip_tables_device_ptr(trxhip_ctx *ctx, void **d_blob)
{
	if (!ctx || !d_blob)
		return TRXHIP_EINVAL;
	*d_blob = ctx->d_tables;
	return TRXHIP_OK;
}

static int pull_common(trxhip_ctx *ctx, const void *d_iq, int cf32, const trxhip_burst_params *d_params,
		       const float *d_ebp_in, trxhip_burst_result *d_results, float *d_soft, size_t n_bursts, int burst_len, int sps,
		       float threshold, float full_scale, int soft_stride, int flags, void *stream)
{
	if (!ctx)
		return TRXHIP_EINVAL;
	if (sps != 1 && sps != 4)
		return TRXHIP_EINVAL;                                  /* sigProcLib.cpp:1740-1741 */
	if (burst_len > TRXHIP_MAX_BURST_LEN || (sps == 4 && burst_len < 624) ||
	    (sps == 1 && (burst_len < 148 || burst_len > 192)))
		return TRXHIP_EINVAL;
	if (d_soft && soft_stride < 1)
		return TRXHIP_EINVAL;
	if (n_bursts == 0)
		return TRXHIP_OK;                                      /* empty batch: nothing to do */
	if (!d_iq || !d_params || !d_results || n_bursts > 0x7fffffffull)   /* 32-bit burst index + grid stride in the kernels */
		return TRXHIP_EINVAL;
	if ((reinterpret_cast<uintptr_t>(d_iq) & 3) != 0)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	if (flags & ~(TRXHIP_FLAG_SLICE | TRXHIP_FLAG_EXACT_DEMOD | TRXHIP_FLAG_IDLE_DUMMY | TRXHIP_FLAG_FEW_NB_SLOTS | TRXHIP_FLAG_DIAG_MASK))
		return TRXHIP_EINVAL;
	const bool few_nb = (flags & TRXHIP_FLAG_FEW_NB_SLOTS) != 0;  /* a hint, not a kernel flag */
	flags &= ~TRXHIP_FLAG_FEW_NB_SLOTS;
	if (ctx->no_unit)
		flags |= TRXHIP_IFLAG_NO_UNIT;
	if (ctx->no_sym)
		flags |= TRXHIP_IFLAG_NO_SYM;
	if (ctx->no_fast)
		flags |= TRXHIP_IFLAG_NO_FAST;
	/* A launch that is being captured into a HIP graph keeps whatever it is handed for as long as the graph lives, and replays
	 * may overlap later launches: it gets neither a pool counter pair (static split) nor a leftover list (general kernel only) --
	 * both are recycled between launches. */
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	const bool capturing = stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap) == hipSuccess &&
			       cap != hipStreamCaptureStatusNone;
	/* a counter pair for this launch (zero: the previous user's last workgroup re-armed it; trx_ctx.h) */
	unsigned *pool = nullptr;
	if (ctx->d_pool && ctx->pool_enabled && !capturing && n_bursts >= (size_t)ctx->n_cu * 128)
		pool = ctx->d_pool + (size_t)(ctx->pool_next.fetch_add(1u, std::memory_order_relaxed) % TRX_POOL_SLOTS) * 16;
	/* The call pullRadioVector() makes for its traffic slots (int16 bursts of 625 samples at 4 SPS, fused demodulator, sliced rows
	 * of 148 soft bits): the normal-burst kernel over the batch, then the general kernel over whatever it left on its list. */
	if (ctx->nb_enabled && !few_nb && !capturing && !cf32 && !d_ebp_in && sps == 4 && burst_len == 625 && d_soft && soft_stride == 148 &&
	    flags == TRXHIP_FLAG_SLICE) {
		std::lock_guard<std::mutex> lk(ctx->redo_mu);
		if (ctx->split_backoff) {
			ctx->split_backoff--;                                   /* a recent batch left too much: the general kernel alone */
			return trx_launch_pull(pool, d_iq, cf32, d_params, d_results, d_soft, ctx->d_tables, d_ebp_in, n_bursts, burst_len, sps,
					       threshold, full_scale, soft_stride, flags, ctx->n_cu, static_cast<hipStream_t>(stream));
		}
		trxhip_ctx::redo_slot &sl = ctx->redo[ctx->redo_next++ % TRX_REDO_SLOTS];
		bool ok = true;
		if (sl.busy) {
			ok = hipEventSynchronize(sl.ev) == hipSuccess;
			sl.busy = 0;
			if (ok && sl.h_left && sl.n_last && (size_t)*reinterpret_cast<volatile unsigned *>(sl.h_left) * 32 > sl.n_last)
				ctx->split_backoff = ctx->no_backoff ? 0 : 63;
		}
		if (ok && !sl.h_left)
			ok = hipHostMalloc(reinterpret_cast<void **>(&sl.h_left), 64, hipHostMallocDefault) == hipSuccess;
		if (ok) {
			*sl.h_left = 0u;
			sl.n_last = n_bursts;
		}
		if (ok && !sl.ev)
			ok = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) == hipSuccess;
		if (ok && sl.cap < n_bursts) {
			if (sl.d) (void)hipFree(sl.d);
			sl.d = nullptr;
			sl.cap = 0;
			const size_t cap_new = n_bursts < 65536 ? 65536 : n_bursts;
			/* header words + one flag byte per burst, padded to whole chunks of 256; all zero between launches */
			const size_t bytes = TRX_REDO_HDR_WORDS * sizeof(unsigned) + ((cap_new + 255) & ~(size_t)255);
			/* (the memset runs on the null stream: drained before a kernel on the caller's -- possibly non-blocking -- stream
			 * may touch the buffer; this is the allocation path, a handful of calls per context) */
			ok = hipMalloc(reinterpret_cast<void **>(&sl.d), bytes) == hipSuccess && hipMemset(sl.d, 0, bytes) == hipSuccess &&
			     hipDeviceSynchronize() == hipSuccess;
			if (ok)
				sl.cap = cap_new;
			else if (sl.d) {
				(void)hipFree(sl.d);
				sl.d = nullptr;
			}
		}
		if (ok) {
			const int rc = trx_launch_pull4_nb(pool, d_iq, d_params, d_results, d_soft, ctx->d_tables, n_bursts, threshold, full_scale,
							   ctx->n_cu, sl.d, sl.h_left, static_cast<hipStream_t>(stream));
			if (rc == 0 && hipEventRecord(sl.ev, static_cast<hipStream_t>(stream)) == hipSuccess)
				sl.busy = 1;
			else if (rc == 0)
				(void)hipStreamSynchronize(static_cast<hipStream_t>(stream));   /* no event: the slot is free once the stream has drained */
			return rc;
		}
		/* (no list buffer: fall through to the general kernel) */
	}
	return trx_launch_pull(pool, d_iq, cf32, d_params, d_results, d_soft, ctx->d_tables, d_ebp_in, n_bursts, burst_len, sps,
			       threshold, full_scale, soft_stride, flags, ctx->n_cu, static_cast<hipStream_t>(stream));
}

extern "C" int trx_launch_diversity_select(const int16_t *d_iq_paths, size_t n_bursts, int n_paths, int burst_len, int sps,
					   int16_t *d_iq_sel, float *d_avg_energy, uint8_t *d_path, hipStream_t stream);
extern "C" int trx_launch_diversity_power(trxhip_burst_result *d_res, const trxhip_burst_params *d_params, const float *d_avg_energy,
					  size_t n_bursts, float full_scale, hipStream_t stream);

int trxhip_select_diversity_batch(trxhip_ctx *ctx, const int16_t *d_iq_paths, size_t n_bursts, int n_paths, int burst_len, int sps,
				  int16_t *d_iq_sel, float *d_avg_energy, uint8_t *d_path, void *stream)
{
	if (!ctx || n_paths < 1 || n_paths > 8 || (sps != 1 && sps != 4) || burst_len < 1 || burst_len > TRXHIP_MAX_BURST_LEN)
		return TRXHIP_EINVAL;
	/* energyDetect(path, 20 * sps) reads samples 0, 4, 8, ... of the path (sigProcLib.cpp:1576-1584, window clamped to the
	 * vector's size): the last one, 4 * (window - 1), must lie inside the path -- 317 samples at 4 SPS, 77 at 1 SPS */
	{
		const int window = 20 * sps < burst_len ? 20 * sps : burst_len;
		if (4 * (window - 1) >= burst_len)
			return TRXHIP_EINVAL;
	}
	if (n_bursts == 0)
		return TRXHIP_OK;
	if (!d_iq_paths || !d_iq_sel || !d_avg_energy || (reinterpret_cast<uintptr_t>(d_iq_paths) & 3) ||
	    (reinterpret_cast<uintptr_t>(d_iq_sel) & 3))
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_diversity_select(d_iq_paths, n_bursts, n_paths, burst_len, sps, d_iq_sel, d_avg_energy, d_path,
					   static_cast<hipStream_t>(stream));
}

int trxhip_apply_diversity_power(trxhip_ctx *ctx, trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
				 const float *d_avg_energy, size_t n_bursts, float full_scale, void *stream)
{
	if (!ctx || !(full_scale > 0.0f))
		return TRXHIP_EINVAL;
	if (n_bursts == 0)
		return TRXHIP_OK;
	if (!d_results || !d_params || !d_avg_energy)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_diversity_power(d_results, d_params, d_avg_energy, n_bursts, full_scale, static_cast<hipStream_t>(stream));
}

int trxhip_detect_demod_batch(trxhip_ctx *ctx, const int16_t *d_iq, const trxhip_burst_params *d_params,
			      trxhip_burst_result *d_results, float *d_soft, size_t n_bursts, int burst_len, int sps,
			      float threshold, float full_scale, int soft_stride, int flags, void *stream)
{
	return pull_common(ctx, d_iq, 0, d_params, nullptr, d_results, d_soft, n_bursts, burst_len, sps, threshold, full_scale,
			   soft_stride, flags, stream);
}

int trxhip_detect_demod_batch_cf32(trxhip_ctx *ctx, const float *d_iq, const trxhip_burst_params *d_params,
				   trxhip_burst_result *d_results, float *d_soft, size_t n_bursts, int burst_len,
				   int sps, float threshold, float full_scale, int soft_stride, int flags, void *stream)
{
	return pull_common(ctx, d_iq, 1, d_params, nullptr, d_results, d_soft, n_bursts, burst_len, sps, threshold, full_scale,
			   soft_stride, flags, stream);
}

int trxhip_demod_batch_cf32(trxhip_ctx *ctx, const float *d_iq, const trxhip_burst_params *d_params,
			    const float *d_ebp, trxhip_burst_result *d_results, float *d_soft, size_t n_bursts,
			    int burst_len, int sps, int soft_stride, int flags, void *stream)
{
	if (n_bursts > 0 && (!d_ebp || (reinterpret_cast<uintptr_t>(d_ebp) & 15) != 0))
		return TRXHIP_EINVAL;
	return pull_common(ctx, d_iq, 1, d_params, d_ebp, d_results, d_soft, n_bursts, burst_len, sps,
			   TRXHIP_BURST_THRESH, 1.0f, soft_stride, flags, stream);
}

int trxhip_pack_trxd_batch(trxhip_ctx *ctx, const trxhip_burst_result *d_results, const float *d_soft_sliced,
			   int soft_stride, uint8_t *d_pkt, size_t n_bursts, float rssi_offset, void *stream)
{
	if (!ctx || !d_results || !d_soft_sliced || !d_pkt || soft_stride < 148)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_pack_trxd(d_results, d_soft_sliced, soft_stride, d_pkt, n_bursts, rssi_offset,
				    static_cast<hipStream_t>(stream));
}

int trxhip_pack_trxd_wire_batch(trxhip_ctx *ctx, const trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
				const float *d_soft_sliced, int soft_stride, const trxhip_trxd_meta *d_meta,
				uint8_t *d_pkt, int pkt_stride, uint16_t *d_pkt_len, size_t n_bursts, float rssi_offset,
				void *stream)
{
	if (!ctx || soft_stride < 148 || pkt_stride < 160 || (pkt_stride & 3))
		return TRXHIP_EINVAL;
	if (n_bursts == 0)
		return TRXHIP_OK;
	if (!d_results || !d_params || !d_soft_sliced || !d_meta || !d_pkt || !d_pkt_len || n_bursts > 0x7fffffffull ||
	    (reinterpret_cast<uintptr_t>(d_pkt) & 3) != 0)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_pack_trxd_wire(d_results, d_params, d_soft_sliced, soft_stride, d_meta, d_pkt, pkt_stride, d_pkt_len,
					 n_bursts, rssi_offset, static_cast<hipStream_t>(stream), nullptr);
}

static int conv_common(trxhip_ctx *ctx, const float *d_x, int x_len, const float *d_h, int h_len, int h_complex,
		       float *d_y, int y_len, int start, int len, size_t n_vec, void *stream)
{
	if (!ctx || !d_x || !d_h || !d_y)
		return TRXHIP_EINVAL;
	/* bounds_check(), arch/common/convolve_base.c:88-105, plus the head-room rule */
	if (x_len < 1 || h_len < 1 || y_len < 1 || len < 1 || h_len > 256)
		return TRXHIP_EINVAL;
	if (start + len > x_len || len > y_len || x_len < h_len || start < h_len - 1)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_convolve(d_x, x_len, d_h, h_len, h_complex, d_y, y_len, start, len, n_vec,
				   static_cast<hipStream_t>(stream));
}

int trxhip_convolve_real_batch(trxhip_ctx *ctx, const float *d_x, int x_len, const float *d_h, int h_len,
			       float *d_y, int y_len, int start, int len, size_t n_vec, void *stream)
{
	return conv_common(ctx, d_x, x_len, d_h, h_len, 0, d_y, y_len, start, len, n_vec, stream);
}

int trxhip_convolve_complex_batch(trxhip_ctx *ctx, const float *d_x, int x_len, const float *d_h, int h_len,
				  float *d_y, int y_len, int start, int len, size_t n_vec, void *stream)
{
	return conv_common(ctx, d_x, x_len, d_h, h_len, 1, d_y, y_len, start, len, n_vec, stream);
}

int trxhip_convert_short_float(trxhip_ctx *ctx, float *d_out, const int16_t *d_in, size_t len, void *stream)
{
	if (!ctx || !d_out || !d_in)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_convert_short_float(d_out, d_in, len, static_cast<hipStream_t>(stream));
}

int trxhip_convert_float_short(trxhip_ctx *ctx, int16_t *d_out, const float *d_in, float scale, size_t len, void *stream)
{
	if (!ctx || (len && (!d_out || !d_in)))
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_convert_float_short(d_out, d_in, scale, len, static_cast<hipStream_t>(stream));
}

int trxhip_dft_batch(trxhip_ctx *ctx, const float *d_in, float *d_out, int m, size_t howmany, size_t istride, size_t ostride,
		     int reverse, void *stream)
{
	if (!ctx || m < 1 || m > 4096 || istride < howmany || ostride < howmany)
		return TRXHIP_EINVAL;
	if (howmany == 0)
		return TRXHIP_OK;
	if (!d_in || !d_out || d_in == d_out)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_dft_strided(d_in, d_out, m, howmany, istride, ostride, reverse, static_cast<hipStream_t>(stream));
}

int trxhip_energy_detect_batch_cf32(trxhip_ctx *ctx, const float *d_iq, size_t n_bursts, int burst_len,
				    unsigned window, float *d_energy, void *stream)
{
	if (!ctx || burst_len < 1)
		return TRXHIP_EINVAL;
	if (n_bursts == 0)
		return TRXHIP_OK;
	if (!d_iq || !d_energy)
		return TRXHIP_EINVAL;
	if (window > 0 && 4 * (size_t)(window > (unsigned)burst_len ? (unsigned)burst_len : window) - 3 > (size_t)burst_len)
		return TRXHIP_EINVAL;                  /* the reference would read past the burst (sigProcLib.cpp:1580-1583) */
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_energy_detect(d_iq, n_bursts, burst_len, window, d_energy, static_cast<hipStream_t>(stream));
}

int trxhip_detect_sch_batch_cf32(trxhip_ctx *ctx, const float *d_iq, trxhip_burst_result *d_results, size_t n_bufs,
				 size_t buf_len, int sps, int state, float threshold, void *stream)
{
	if (!ctx || (sps != 1 && sps != 4))
		return TRXHIP_EINVAL;                  /* sigProcLib.cpp:1814-1815 */
	/* window of the search (:1817-1838) */
	int target = 3 + 39 + 64, head, tail;
	switch (state) {
	case TRXHIP_SCH_DETECT_NARROW: head = 4; tail = 4; break;
	case TRXHIP_SCH_DETECT_BUFFER: target = 1; head = 0; tail = (12 * 8 * 625) / 4; break;
	case TRXHIP_SCH_DETECT_FULL:
	default: head = target - 1; tail = 39 + 3 + 9; break;
	}
	const int start = (target - head) * 1 - 1, len = (head + tail) * 1;
	if (buf_len < 4 * (size_t)len)
		return TRXHIP_EINVAL;
	if (n_bufs == 0)
		return TRXHIP_OK;
	if (!d_iq || !d_results || n_bufs > 0x7fffffffull)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	const int toa_sub = (state == TRXHIP_SCH_DETECT_BUFFER) ? 3 + 39 + 64 : head;      /* :1853-1858 */
	return trx_launch_sch_detect(d_iq, buf_len, d_results, ctx->d_tables, n_bufs, len, start, toa_sub, threshold,
				     ctx->sch_unit, static_cast<hipStream_t>(stream));
}

int trxhip_delay_vector_batch_cf32(trxhip_ctx *ctx, const float *d_in, float *d_out, const float *d_delays, size_t n_vec,
				   int len, void *stream)
{
	if (!ctx || len < 0)
		return TRXHIP_EINVAL;
	if (n_vec == 0 || len == 0)
		return TRXHIP_OK;
	if (!d_in || !d_out || !d_delays || d_in == d_out)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_delay_vector(d_in, d_out, d_delays, ctx->d_tables, n_vec, len, static_cast<hipStream_t>(stream));
}

int trxhip_scale_vector_cf32(trxhip_ctx *ctx, float *d_x, size_t len, float scale_re, float scale_im, void *stream)
{
	if (!ctx || (len && !d_x))
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_scale_vector(d_x, len, scale_re, scale_im, static_cast<hipStream_t>(stream));
}

int trxhip_demod_va_batch_cf32(trxhip_ctx *ctx, const float *d_iq, const trxhip_burst_params *d_params,
			       const trxhip_burst_result *d_detected, float *d_soft, int32_t *d_starts, size_t n_bursts,
			       int burst_len, float scale, int soft_stride, int flags, void *stream)
{
	if (!ctx || burst_len < 1 || burst_len > TRXHIP_MAX_BURST_LEN || soft_stride < 148)
		return TRXHIP_EINVAL;
	if (n_bursts == 0)
		return TRXHIP_OK;
	if (!d_iq || !d_params || !d_soft || n_bursts > 0x7fffffffull)
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_va_demod(d_iq, d_params, d_detected, d_soft, d_starts, n_bursts, burst_len, scale, soft_stride, flags,
				   static_cast<hipStream_t>(stream));
}

int trxhip_vector_slicer(trxhip_ctx *ctx, float *d_dest, const float *d_src, size_t len, void *stream)
{
	if (!ctx || (len && (!d_dest || !d_src)))
		return TRXHIP_EINVAL;
	if (with_device(ctx))
		return TRXHIP_EIO;
	return trx_launch_vector_slicer(d_dest, d_src, len, static_cast<hipStream_t>(stream));
}

int trxhip_channelize_batch(trxhip_ctx *ctx, const int16_t *d_in, float *d_out, size_t n_blocks, int m,
			    int block_len, int h_len, void *stream)
{
	if (!ctx || !d_in || !d_out || block_len < 1)
		return TRXHIP_EINVAL;
	if (m != 4 || h_len != 16)
		return TRXHIP_ENOTSUP;                 /* the reference instantiates Channelizer(4, 192, 16) only */
	if ((reinterpret_cast<uintptr_t>(d_in) & 15) != 0)
		return TRXHIP_EINVAL;                  /* one 16-byte load per time step */
	if (with_device(ctx))
		return TRXHIP_EIO;
	const size_t n_total = n_blocks * (size_t)block_len;
	return trx_launch_channelize(d_in, d_out, n_total, n_total, ctx->d_tables, nullptr, static_cast<hipStream_t>(stream));
}

int trxhip_resample_batch(trxhip_ctx *ctx, const float *d_in, float *d_out, size_t n_in, int p, int q,
			  size_t n_chan, size_t in_stride, size_t out_stride, void *stream)
{
	if (!ctx || !d_in || !d_out)
		return TRXHIP_EINVAL;
	if (!((p == 65 && q == 48) || (p == 1 && q == 4)))
		return TRXHIP_ENOTSUP;                 /* Resampler(65,48) radioInterfaceMulti.cpp:35-36; (1,4) sigProcLib.cpp:2161 */
	if (n_in % q)
		return TRXHIP_EINVAL;                  /* Resampler.cpp:100-104 */
	if (with_device(ctx))
		return TRXHIP_EIO;
	const float *parts = (p == 65) ? &ctx->d_tables->rs6548_taps[0][0] : &ctx->d_tables->dec_taps[0];
	return trx_launch_resample(d_in, d_out, n_in, p, q, n_chan, in_stride, out_stride, parts, nullptr,
				   static_cast<hipStream_t>(stream));
}

/* ---- streaming Rx front end: RadioInterfaceMulti::pullBuffer (radioInterfaceMulti.cpp:237-314) ---- */
struct trxhip_rx_frontend {
	trxhip_ctx *ctx;
	int block_len, p, q;
	float *d_parts;          /* [p][16] resampler partitions */
	void *d_wide_hist;       /* 15 time steps x 4 int16 IQ samples */
	void *d_chan_hist;       /* [2][4][16] complex64: the fused kernel reads one half and leaves the other (its first and its
	                          * last workgroup run at the same time) */
	int hist_cur;
	float *d_chan;           /* [4][cap] channelizer output scratch */
	size_t cap;
};

int trxhip_rx_frontend_create(trxhip_ctx *ctx, int block_len, int p, int q, trxhip_rx_frontend **out)
{
	if (!ctx || !out || block_len < 16 || p < 1 || q < 1 || p > 128 || q > 3072 || (block_len % q) != 0)
		return TRXHIP_EINVAL;                  /* Resampler::rotate needs whole q-sample groups per block (Resampler.cpp:100-112) */
	if (with_device(ctx))
		return TRXHIP_EIO;
	trxhip_rx_frontend *f = new (std::nothrow) trxhip_rx_frontend();
	if (!f)
		return TRXHIP_ENOMEM;
	f->ctx = ctx; f->block_len = block_len; f->p = p; f->q = q; f->d_chan = nullptr; f->cap = 0; f->hist_cur = 0;
	float *taps = static_cast<float *>(malloc((size_t)p * 16 * sizeof(float)));
	if (!taps) { delete f; return TRXHIP_ENOMEM; }
	trx_polyphase_taps((unsigned)p, (unsigned)q, 16, 1.0f, taps);
	bool ok = hipMalloc((void **)&f->d_parts, (size_t)p * 16 * sizeof(float)) == hipSuccess &&
		  hipMalloc(&f->d_wide_hist, 16 * 16) == hipSuccess && hipMalloc(&f->d_chan_hist, 2 * 4 * 16 * 8) == hipSuccess &&
		  hipMemcpy(f->d_parts, taps, (size_t)p * 16 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
		  hipMemset(f->d_wide_hist, 0, 16 * 16) == hipSuccess && hipMemset(f->d_chan_hist, 0, 2 * 4 * 16 * 8) == hipSuccess;
	free(taps);
	if (!ok) { trxhip_rx_frontend_destroy(f); return TRXHIP_ENOMEM; }
	*out = f;
	return TRXHIP_OK;
}

void trxhip_rx_frontend_destroy(trxhip_rx_frontend *f)
{
	if (!f)
		return;
	if (with_device(f->ctx) == 0) {
		if (f->d_parts) (void)hipFree(f->d_parts);
		if (f->d_wide_hist) (void)hipFree(f->d_wide_hist);
		if (f->d_chan_hist) (void)hipFree(f->d_chan_hist);
		if (f->d_chan) (void)hipFree(f->d_chan);
	}
	delete f;
}

int trxhip_rx_frontend_reset(trxhip_rx_frontend *f, void *stream)
{
	if (!f || with_device(f->ctx))
		return TRXHIP_EINVAL;
	if (hipMemsetAsync(f->d_wide_hist, 0, 16 * 16, static_cast<hipStream_t>(stream)) != hipSuccess ||
	    hipMemsetAsync(f->d_chan_hist, 0, 2 * 4 * 16 * 8, static_cast<hipStream_t>(stream)) != hipSuccess)
		return TRXHIP_EIO;
	return TRXHIP_OK;
}

int trxhip_rx_frontend_seed(trxhip_rx_frontend *f, const int16_t *d_wide_prev, size_t n_blocks_prev, void *stream)
{
	if (!f)
		return TRXHIP_EINVAL;
	int rc = trxhip_rx_frontend_reset(f, stream);
	if (rc || n_blocks_prev == 0)
		return rc;
	if (!d_wide_prev)
		return TRXHIP_EINVAL;
	/* run the preceding blocks through the very same two launches (their output is discarded): what they leave in
	 * d_wide_hist / d_chan_hist is the state of a stream processed up to here */
	const size_t n_out = n_blocks_prev * (size_t)f->block_len / f->q * f->p;
	float *scratch = nullptr;
	if (hipMalloc((void **)&scratch, 4 * n_out * 8) != hipSuccess)
		return TRXHIP_ENOMEM;
	rc = trxhip_rx_frontend_pull(f, d_wide_prev, n_blocks_prev, scratch, n_out, stream);
	if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess && rc == TRXHIP_OK)
		rc = TRXHIP_EIO;
	(void)hipFree(scratch);
	return rc;
}

int trxhip_rx_frontend_pull(trxhip_rx_frontend *f, const int16_t *d_wide, size_t n_blocks, float *d_out,
			    size_t out_stride, void *stream)
{
	if (!f || !d_wide || !d_out || (reinterpret_cast<uintptr_t>(d_wide) & 15) != 0)
		return TRXHIP_EINVAL;
	if (n_blocks == 0)
		return TRXHIP_OK;
	if (with_device(f->ctx))
		return TRXHIP_EIO;
	const size_t n_total = n_blocks * (size_t)f->block_len;
	if (out_stride < n_total / f->q * f->p)
		return TRXHIP_EINVAL;
	hipStream_t s = static_cast<hipStream_t>(stream);
	char *const hist = static_cast<char *>(f->d_chan_hist);
	void *const hist_in = hist + (size_t)f->hist_cur * 4 * 16 * 8, *const hist_out = hist + (size_t)(f->hist_cur ^ 1) * 4 * 16 * 8;
	/* one pass, the channel-rate streams stay on the chip (trx_aux_kernels.hip, frontend_fused_kernel) ... */
	static const bool no_fused = getenv("TRXHIP_NO_FUSED_FRONTEND") != nullptr;
	int rc = no_fused ? 1 : trx_launch_frontend_fused(d_wide, d_out, n_total, f->p, f->q, out_stride, f->d_parts, f->ctx->d_tables,
							  f->d_wide_hist, hist_in, hist_out, s);
	if (rc == 0)
		f->hist_cur ^= 1;
	if (rc != 1)
		return rc;
	/* ... or, for a geometry that does not fit its tiles, the two kernels with the channel streams in a scratch buffer */
	if (n_total > f->cap) {
		if (f->d_chan) (void)hipFree(f->d_chan);
		f->d_chan = nullptr;
		if (hipMalloc((void **)&f->d_chan, 4 * n_total * 8) != hipSuccess) { f->cap = 0; return TRXHIP_ENOMEM; }
		f->cap = n_total;
	}
	rc = trx_launch_channelize(d_wide, f->d_chan, n_total, f->cap, f->ctx->d_tables, f->d_wide_hist, s);
	if (rc)
		return rc;
	return trx_launch_resample(f->d_chan, d_out, n_total, f->p, f->q, 4, f->cap, out_stride, f->d_parts, hist_in, s);
}

}  // extern "C"
