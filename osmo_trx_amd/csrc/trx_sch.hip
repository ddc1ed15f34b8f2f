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
// Sums follow the reference's order (taps k ascending, -ffp-contract=off): rc and TOA are bit-exact.
#include "trx_device.h"

#define SCH_THREADS 256
#define SCH_N 64                       // gSCHSequence length (sigProcLib.cpp:1467-1527)
#define SCH_WIN (2 * TRX_CZ_PAD + 1)   // correlation window kept for the tail: peak +- 12

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

__global__ void __launch_bounds__(SCH_THREADS)
sch_detect_kernel(const c32 *__restrict__ iq, size_t buf_stride, trxhip_burst_result *__restrict__ results,
		  const trx_tables *__restrict__ tab, int len, int start, int toa_sub, float thresh)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *sincv = reinterpret_cast<float *>(smem);                 // [4128] swizzled sinc LUT
	c32 *taps = reinterpret_cast<c32 *>(sincv + TRX_SINCV_LDS);    // [64]
	float *hdr = reinterpret_cast<float *>(taps + SCH_N);          // [8]
	float *gdec = hdr + 8;                                         // [16]
	float *red_v = gdec + 16;                                      // [256]
	int *red_i = reinterpret_cast<int *>(red_v + SCH_THREADS);     // [256]
	c32 *win = reinterpret_cast<c32 *>(red_i + SCH_THREADS);       // [SCH_WIN + 1]
	c32 *dec = win + SCH_WIN + 1;                                  // [len]

	const int tid = threadIdx.x;
	const c32 *x = iq + (size_t)blockIdx.x * buf_stride;
	const trx_seq *sq = &tab->seq[TRX_SEQ_SCH];

	for (int i = tid; i < TRX_SINCV_LDS; i += SCH_THREADS)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	if (tid < SCH_N)
		taps[tid] = make_float2(sq->taps[tid].re, sq->taps[tid].im);
	if (tid < 8)
		hdr[tid] = reinterpret_cast<const float *>(&sq->gain)[tid];
	if (tid < 16)
		gdec[tid] = tab->dec_taps[tid];
	__syncthreads();

	// ---- downsampleBurst(burst, 4*len, len) (:1587-1601, :1841): dec[i] = sum_k X(4i - 15 + k) * g[k], X = 0 for n < 0
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
		dec[i] = make_float2(yr, yi);
	}
	__syncthreads();

	// ---- correlate + fastPeakDetect: per-thread first strict maximum over ascending i, then block arg-max
	float best = 0.0f;
	int bidx = -1;
	for (int i = tid; i < len; i += SCH_THREADS) {
		const float v = norm2(sch_corr_at(dec, len, taps, i, start));
		if (v > best) { best = v; bidx = i; }
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
	if (tid >= WAVE)
		return;

	// ---- wave 0: window of the correlation around the peak, then the shared tail
	const int lane = tid;
	bidx = uni(red_i[0]);
	int rc = 0;
	float toa = 0.0f, ci = 0.0f;
	c32 amp = make_float2(0.0f, 0.0f);
	if (bidx >= 0) {
		if (lane < SCH_WIN) {
			const int g = bidx - TRX_CZ_PAD + lane;
			win[lane] = (g >= 0 && g < len) ? sch_corr_at(dec, len, taps, g, start) : make_float2(0.0f, 0.0f);
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
		reinterpret_cast<uint32_t *>(results + blockIdx.x)[lane] = word;
	}
}

extern "C" int trx_launch_sch_detect(const float *d_iq, size_t buf_stride, trxhip_burst_result *d_results,
				     const trx_tables *d_tab, size_t n_bufs, int len, int start, int toa_sub, float thresh,
				     hipStream_t stream)
{
	if (n_bufs == 0)
		return 0;
	const size_t lds = (size_t)(TRX_SINCV_LDS + 8 + 16 + SCH_THREADS) * sizeof(float) + SCH_THREADS * sizeof(int) +
			   (size_t)(SCH_N + SCH_WIN + 1 + len) * sizeof(c32);
	if (lds > 160 * 1024)
		return TRXHIP_EINVAL;
	TRX_ARM_DYNAMIC_LDS(sch_detect_kernel);
	hipLaunchKernelGGL(sch_detect_kernel, dim3((unsigned)n_bufs), dim3(SCH_THREADS), lds, stream,
			   reinterpret_cast<const c32 *>(d_iq), buf_stride, d_results, d_tab, len, start, toa_sub, thresh);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
