// trx_sch.hip -- detectSCHBurst() (Transceiver52M/sigProcLib.cpp:1805-1861) for gfx950.
//
// The MS-side synchronisation search: the first 4*len samples of a buffer are decimated by 4, correlated with
// the 64-symbol SCH extended training sequence over `len` positions, and the usual detectBurst() tail
// (sigProcLib.cpp:1683-1708) runs on the strongest position.  len is 156 (SCH_DETECT_FULL), 8 (NARROW) or
// 15000 (SCH_DETECT_BUFFER: 12 frames).
//
// Mapping: ONE WORKGROUP (256 threads) PER BUFFER.
//   * the decimated signal lives in LDS (len x 8 B: 1.2 KB for a burst, 117 KB for the 12-frame search --
//     that is why the workgroup owns a whole CU); it is read back 64x by the correlation
//   * the correlation is never stored: pass 1 keeps only each thread's first strict maximum of |corr|^2,
//     a block arg-max (ties -> lowest index) reproduces fastPeakDetect()'s sequential scan (:1120-1139);
//     wave 0 then recomputes the 25 correlation values the tail can touch (peak +- 12) into a window
//   * the tail (edge gate, computePeakRatio, speculative TOA bisection, computeCI) is detect_tail() from
//     trx_device.h, the same code the burst kernels run
//   * round 4: the 64-tap correlation is the addition-only corr_unit() form of the burst kernels (trx_device.h: every tap of
//     the SCH sequence is +-1 rotated by k pi/2 with an fp64 residue <= 7e-14, so under the component-ratio guard each tap
//     contributes +-x0 / +-x1 exactly: ONE v_pk_add_f32 per tap where the multiplying form spends eight instructions and two
//     range checks); the decimated signal sits between zero pads (64 in front, 128 behind) instead of being range-checked
//     per tap.  A buffer with a sample that fails the guard, or tables without the structure, take the multiplying form.
// Sums follow the reference's order (taps k ascending, -ffp-contract=off): rc and TOA are bit-exact.
#include "trx_device.h"

#define SCH_MAX_THREADS 1024
#define SCH_N 64                       // gSCHSequence length (sigProcLib.cpp:1467-1527)
#define SCH_WIN (2 * TRX_CZ_PAD + 1)   // correlation window kept for the tail: peak +- 12
#define SCH_PAD_F 64                   // zero samples in front of the decimated signal (taps reach 63 back)
#define SCH_PAD_B 128                  // ... and behind it (NARROW correlates up to start = 101 positions past its 8 samples)

__device__ __forceinline__ c32 sch_corr_at(const c32 *dec, int len, const c32 *taps, int i, int start)
{
	// corr[i] = sum_k DEC(i + start - 63 + k) * seq[k], DEC = 0 outside [0, len)   (:1674, convolve CUSTOM :327-334)
	float yr = 0.0f, yi = 0.0f;
	const int base = i + start - (SCH_N - 1);
#pragma unroll 8
	for (int k = 0; k < SCH_N; k++) {
		const int j = base + k;
		const c32 x = (j >= 0 && j < len) ? dec[j] : make_float2(0.0f, 0.0f);
		const c32 h = taps[k];
		yr += x.x * h.x - x.y * h.y;
		yi += x.x * h.y + x.y * h.x;
	}
	return make_float2(yr, yi);
}

// the same value from the padded signal with additions only (corr_unit(), trx_device.h); dec[-64 .. len + 128) readable
__device__ __forceinline__ c32 sch_corr_unit(const c32 *dec, int i, int start)
{
	const trx_v2f z = { 0.0f, 0.0f };
	const trx_v2f a = UnitCorr<TRX_UNIT_NEG_SCH, SCH_N>::run(z, dec + (i + start - (SCH_N - 1)));
	return make_float2(a.x, a.y);
}

// the compiled-in sign pattern against what the table generator derived from the taps (host side)
extern "C" int trx_unit_mask_sch_match(const trx_tables *t)
{
	return ((t->unit_ok >> TRX_SEQ_SCH) & 1u) && t->unit_neg[TRX_SEQ_SCH] == TRX_UNIT_NEG_SCH;
}

// Round 4: a workgroup is PERSISTENT -- it stages the tables (16.5 KB of sinc LUT) once and walks buffers blockIdx.x,
// blockIdx.x + gridDim.x, ... (the FULL search's 16384 small buffers spent most of their time re-staging) -- and its size follows
// the search: 256 threads for the burst-sized searches, 1024 for the 12-frame one (15000 positions: 16 waves per CU instead of 4).
__global__ void __launch_bounds__(SCH_MAX_THREADS)
sch_detect_kernel(const c32 *__restrict__ iq, size_t buf_stride, trxhip_burst_result *__restrict__ results,
		  const trx_tables *__restrict__ tab, int len, int start, int toa_sub, float thresh, int unit_tables, unsigned n_bufs)
{
	const int SCH_THREADS = (int)blockDim.x;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *sincv = reinterpret_cast<float *>(smem);                 // [4128] swizzled sinc LUT
	c32 *taps = reinterpret_cast<c32 *>(sincv + TRX_SINCV_LDS);    // [64]
	float *hdr = reinterpret_cast<float *>(taps + SCH_N);          // [8]
	float *gdec = hdr + 8;                                         // [16]
	float *red_v = gdec + 16;                                      // [256]
	int *red_i = reinterpret_cast<int *>(red_v + SCH_THREADS);     // [256]
	int *guard = red_i + SCH_THREADS;                              // [2] (one used): a sample failed the unit-correlation guard
	c32 *win = reinterpret_cast<c32 *>(guard + 2);                 // [SCH_WIN + 1]
	c32 *dec = win + SCH_WIN + 1 + SCH_PAD_F;                      // [-64 .. len + 128): zero pads either side

	const int tid = threadIdx.x;
	const trx_seq *sq = &tab->seq[TRX_SEQ_SCH];

	for (int i = tid; i < TRX_SINCV_LDS; i += SCH_THREADS)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	if (tid < SCH_N)
		taps[tid] = make_float2(sq->taps[tid].re, sq->taps[tid].im);
	if (tid < 8)
		hdr[tid] = reinterpret_cast<const float *>(&sq->gain)[tid];
	if (tid < 16)
		gdec[tid] = tab->dec_taps[tid];
	if (tid < SCH_PAD_F)
		dec[tid - SCH_PAD_F] = make_float2(0.0f, 0.0f);
	if (tid < SCH_PAD_B)
		dec[len + tid] = make_float2(0.0f, 0.0f);
	__syncthreads();

	for (unsigned buf = blockIdx.x; buf < n_bufs; buf += gridDim.x) {
	const c32 *x = iq + (size_t)buf * buf_stride;
	if (tid == 0)
		guard[0] = 0;
	__syncthreads();                                               // (also: the previous buffer's tail has finished with dec[] / win[])

	// ---- downsampleBurst(burst, 4*len, len) (:1587-1601, :1841): dec[i] = sum_k X(4i - 15 + k) * g[k], X = 0 for n < 0
	bool bad = false;                                              // guard of the addition-only correlation (unit_unsafe())
	for (int i = tid; i < len; i += SCH_THREADS) {
		float yr = 0.0f, yi = 0.0f;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const int j = 4 * i - 15 + k;
			const c32 v = (j >= 0) ? x[j] : make_float2(0.0f, 0.0f);   // j < 4*len always
			const float g = gdec[k];
			yr += v.x * g;
			yi += v.y * g;
		}
		const c32 y = make_float2(yr, yi);
		dec[i] = y;
		bad |= unit_unsafe(y);
	}
	if (bad)
		guard[0] = 1;                                              // (any writer will do; __syncthreads_or() would bring static LDS
	__syncthreads();                                               //  next to the 160 KB dynamic opt-in)
	const bool unit = unit_tables && guard[0] == 0;                // workgroup-uniform

	// ---- correlate + fastPeakDetect: per-thread first strict maximum over ascending i, then block arg-max
	float best = 0.0f;
	int bidx = -1;
	if (unit) {
		for (int i = tid; i < len; i += SCH_THREADS) {
			const float v = norm2(sch_corr_unit(dec, i, start));
			if (v > best) { best = v; bidx = i; }
		}
	} else {
		for (int i = tid; i < len; i += SCH_THREADS) {
			const float v = norm2(sch_corr_at(dec, len, taps, i, start));
			if (v > best) { best = v; bidx = i; }
		}
	}
	red_v[tid] = best;
	red_i[tid] = bidx;
	__syncthreads();
	for (int s = SCH_THREADS / 2; s > 0; s >>= 1) {
		if (tid < s) {
			const float v2 = red_v[tid + s];
			const int i2 = red_i[tid + s];
			const float v1 = red_v[tid];
			const int i1 = red_i[tid];
			// the sequential scan keeps the lowest index among equal maxima; idx -1 = "no value above 0"
			if (v2 > v1 || (v2 == v1 && i2 >= 0 && (i1 < 0 || i2 < i1))) { red_v[tid] = v2; red_i[tid] = i2; }
		}
		__syncthreads();
	}
	if (tid < WAVE) {
	// ---- wave 0: window of the correlation around the peak, then the shared tail
	const int lane = tid;
	bidx = uni(red_i[0]);
	int rc = 0;
	float toa = 0.0f, ci = 0.0f;
	c32 amp = make_float2(0.0f, 0.0f);
	if (bidx >= 0) {
		if (lane < SCH_WIN) {
			const int g = bidx - TRX_CZ_PAD + lane;
			win[lane] = (g >= 0 && g < len) ? (unit ? sch_corr_unit(dec, g, start) : sch_corr_at(dec, len, taps, g, start))
							 : make_float2(0.0f, 0.0f);
		}
		wave_sync();
		const PeakConst pkc = peak_const(lane);
#ifdef TRX_DIAG
		unsigned long long diag_acc[24] = {0}, diag_prev = 0;
#endif
		rc = detect_tail(dec, len, win - (bidx - TRX_CZ_PAD), hdr, SCH_N, thresh, start, len, bidx, sincv, pkc, lane,
				       &toa, &amp, &ci, 0 DIAG_PASS);
	}
	if (lane < 8) {
		const bool det = rc > 0;
		// :1846-1858: on a miss amp = toa = 0; on a hit toa -= head (or 3+39+64 for the buffer search)
		uint32_t word = det ? 1u : 0u;                           // detectBurst()'s rc (:1842)
		word = (lane == 1) ? __float_as_uint(det ? toa - (float)toa_sub : 0.0f) : word;
		word = (lane == 2) ? __float_as_uint(det ? amp.x : 0.0f) : word;
		word = (lane == 3) ? __float_as_uint(det ? amp.y : 0.0f) : word;
		word = (lane == 4) ? __float_as_uint(det ? ci : 0.0f) : word;
		word = (lane == 5 || lane == 6) ? 0u : word;
		word = (lane == 7) ? ((uint32_t)(det ? 0 : 1) << 16) | ((uint32_t)(det ? 148 / 4 : 0) << 24) : word;
		reinterpret_cast<uint32_t *>(results + buf)[lane] = word;
	}
	}   // wave 0
	}   // buffers of this workgroup
}

extern "C" int trx_launch_sch_detect(const float *d_iq, size_t buf_stride, trxhip_burst_result *d_results,
				     const trx_tables *d_tab, size_t n_bufs, int len, int start, int toa_sub, float thresh,
				     int unit_tables, hipStream_t stream)
{
	if (n_bufs == 0)
		return 0;
	const int threads = len > 1024 ? SCH_MAX_THREADS : 256;
	const size_t lds = (size_t)(TRX_SINCV_LDS + 8 + 16 + threads) * sizeof(float) + (threads + 2) * sizeof(int) +
			   (size_t)(SCH_N + SCH_WIN + 1 + SCH_PAD_F + len + SCH_PAD_B) * sizeof(c32);
	if (lds > 160 * 1024)
		return TRXHIP_EINVAL;
	TRX_ARM_DYNAMIC_LDS(sch_detect_kernel);
	/* persistent grid: as many workgroups as fit the chip at this LDS size (256 CUs), each walking its share of the buffers */
	size_t per_cu = lds ? (160 * 1024) / lds : 1;
	if (per_cu < 1) per_cu = 1;
	if (per_cu > (size_t)(2048 / threads)) per_cu = (size_t)(2048 / threads);
	size_t grid = 256 * per_cu;
	if (grid > n_bufs) grid = n_bufs;
	hipLaunchKernelGGL(sch_detect_kernel, dim3((unsigned)grid), dim3(threads), lds, stream,
			   reinterpret_cast<const c32 *>(d_iq), buf_stride, d_results, d_tab, len, start, toa_sub, thresh, unit_tables,
			   (unsigned)n_bufs);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
