// trx_aux_kernels.hip -- the kernels either side of the burst hot path (gfx950, wave64):
//   * convert_short_float            arch/common/convert_base.c:27-31 (radioInterface.cpp:344-348)
//   * convolve_real / _complex       arch/common/convolve_base.c:57-85, batched
//   * Channelizer::rotate            Channelizer.cpp:74-99 (M = 4 polyphase bank + 4-point DFT)
//   * Resampler::rotate              Resampler.cpp:131-150 (65/48 and 1/4)
//   * TRXD payload packing           proto_trxd.c:36-66
// All are streaming, HBM-bound kernels: coalesced loads, int16->fp32 fused into the load,
// taps in LDS, sums in the reference's generic-C order (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "trx_tables.h"
#include "../../include/trxhip.h"

typedef float2 c32;

// ------------------------------------------------------------------------------------------------
// int16 -> fp32, no scaling.  4 shorts (8 B) in, 16 B out per thread-iteration.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
convert_short_float_kernel(float *__restrict__ out, const int16_t *__restrict__ in, size_t len)
{
	const size_t nvec = len / 4;
	const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 7) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
	if (aligned) {
		const short4 *in4 = reinterpret_cast<const short4 *>(in);
		float4 *out4 = reinterpret_cast<float4 *>(out);
		for (size_t i = tid; i < nvec; i += stride) {
			const short4 s = in4[i];
			out4[i] = make_float4((float)s.x, (float)s.y, (float)s.z, (float)s.w);
		}
		for (size_t i = nvec * 4 + tid; i < len; i += stride)
			out[i] = (float)in[i];
	} else {
		for (size_t i = tid; i < len; i += stride)
			out[i] = (float)in[i];
	}
}

extern "C" int trx_launch_convert_short_float(float *d_out, const int16_t *d_in, size_t len, hipStream_t stream)
{
	if (len == 0)
		return 0;
	size_t blocks = (len / 4 + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	if (blocks < 1) blocks = 1;
	hipLaunchKernelGGL(convert_short_float_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_out, d_in, len);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// fp32 -> int16 with scaling, the generic-C form: out[i] = (short)(in[i] * scale)  (convert_base.c:20-25: truncation
// toward zero; the SSE path of the reference rounds to nearest and saturates instead, convert_sse_3.c:29-102)
__global__ void __launch_bounds__(256)
convert_float_short_kernel(int16_t *__restrict__ out, const float *__restrict__ in, float scale, size_t len)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
		out[i] = (int16_t)(int)(in[i] * scale);
}

extern "C" int trx_launch_convert_float_short(int16_t *d_out, const float *d_in, float scale, size_t len, hipStream_t stream)
{
	if (len == 0)
		return 0;
	size_t blocks = (len + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(convert_float_short_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_out, d_in, scale, len);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// cxvec_fft() (arch/common/fft.c:55-114): `howmany` independent M-point DFTs laid out as the reference's
// fftwf_plan_many_dft(rank 1, n = m, howmany, in, istride, idist = 1, out, ostride, odist = 1) call does:
// transform t reads in[j * istride + t] and writes out[k * ostride + t].  One thread per transform.
// M = 4 (the only size the reference instantiates: Channelizer / Synthesis) uses exact +-1 / +-j butterflies;
// other M evaluate X[k] = sum_j x[j] w^(jk) directly with twiddles from sincospi (double) -- FFTW is absent here, so
// there is nothing to be bit-compatible with beyond the mathematical definition ("parity unpinned").
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dft_strided_kernel(const c32 *__restrict__ in, c32 *__restrict__ out, int m, size_t howmany, size_t istride, size_t ostride,
		   int reverse)
{
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < howmany; t += (size_t)gridDim.x * blockDim.x) {
		if (m == 4) {
			const c32 y0 = in[t], y1 = in[istride + t], y2 = in[2 * istride + t], y3 = in[3 * istride + t];
			const c32 t1 = make_float2(y0.x + y2.x, y0.y + y2.y);
			const c32 t2 = make_float2(y0.x - y2.x, y0.y - y2.y);
			const c32 t3 = make_float2(y1.x + y3.x, y1.y + y3.y);
			const c32 t4 = make_float2(y1.x - y3.x, y1.y - y3.y);
			const c32 a = make_float2(t2.x + t4.y, t2.y - t4.x);       // t2 - j*t4
			const c32 b = make_float2(t2.x - t4.y, t2.y + t4.x);       // t2 + j*t4
			out[t] = make_float2(t1.x + t3.x, t1.y + t3.y);
			out[ostride + t] = reverse ? b : a;
			out[2 * ostride + t] = make_float2(t1.x - t3.x, t1.y - t3.y);
			out[3 * ostride + t] = reverse ? a : b;
		} else {
			for (int k = 0; k < m; k++) {
				double ar = 0.0, ai = 0.0;
				for (int j = 0; j < m; j++) {
					double sn, cs;
					sincospi(2.0 * (double)((j * k) % m) / (double)m, &sn, &cs);
					if (!reverse) sn = -sn;
					const c32 x = in[(size_t)j * istride + t];
					ar += (double)x.x * cs - (double)x.y * sn;
					ai += (double)x.x * sn + (double)x.y * cs;
				}
				out[(size_t)k * ostride + t] = make_float2((float)ar, (float)ai);
			}
		}
	}
}

extern "C" int trx_launch_dft_strided(const float *d_in, float *d_out, int m, size_t howmany, size_t istride, size_t ostride,
				      int reverse, hipStream_t stream)
{
	if (howmany == 0)
		return 0;
	size_t blocks = (howmany + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(dft_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const c32 *>(d_in),
			   reinterpret_cast<c32 *>(d_out), m, howmany, istride, ostride, reverse);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// batched correlation-form FIR:  y[v][i] = sum_k x[v][i + start - (H-1) + k] * h[k]
// one thread per output sample, taps staged in LDS, sequential k (generic-C order)
// ------------------------------------------------------------------------------------------------
template <bool HCPLX>
__global__ void __launch_bounds__(256)
convolve_kernel(const c32 *__restrict__ x, int x_len, const c32 *__restrict__ h, int h_len,
		c32 *__restrict__ y, int y_len, int start, int len, size_t n_vec)
{
	__shared__ c32 hs[256];
	for (int k = threadIdx.x; k < h_len; k += blockDim.x)
		hs[k] = h[k];
	__syncthreads();
	const size_t total = n_vec * (size_t)len;
	for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
		const size_t v = o / len;
		const int i = (int)(o - v * len);
		const c32 *xp = x + v * (size_t)x_len + (i + start - (h_len - 1));
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < h_len; k++) {
			const c32 xv = xp[k];
			const c32 t = hs[k];
			if (HCPLX) {                                  // mac_cmplx
				yr += xv.x * t.x - xv.y * t.y;
				yi += xv.x * t.y + xv.y * t.x;
			} else {                                      // mac_real: imag of the tap ignored
				yr += xv.x * t.x;
				yi += xv.y * t.x;
			}
		}
		y[v * (size_t)y_len + i] = make_float2(yr, yi);
	}
}

// The same sums with the vector's window staged in LDS (round 3): one workgroup per vector loads x[start - (H-1) ..
// start + len - 1] once, coalesced, instead of H global loads per output; taps are wave-uniform (scalar loads).
// Used when the window fits 48 KB; the form above otherwise.
template <bool HCPLX>
__global__ void __launch_bounds__(256)
convolve_lds_kernel(const c32 *__restrict__ x, int x_len, const c32 *__restrict__ h, int h_len,
		    c32 *__restrict__ y, int y_len, int start, int len)
{
	extern __shared__ __attribute__((aligned(16))) char cv_smem[];
	c32 *xs = reinterpret_cast<c32 *>(cv_smem);                          // xs[j] = x[v][start - (H-1) + j]
	const size_t v = blockIdx.x;
	const c32 *xv0 = x + v * (size_t)x_len + (start - (h_len - 1));
	const int span = len + h_len - 1;
	for (int j = threadIdx.x; j < span; j += blockDim.x)
		xs[j] = xv0[j];
	__syncthreads();
	for (int i = threadIdx.x; i < len; i += blockDim.x) {
		const c32 *xp = xs + i;
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < h_len; k++) {
			const c32 xv = xp[k];
			const c32 t = h[k];
			if (HCPLX) {                                  // mac_cmplx
				yr += xv.x * t.x - xv.y * t.y;
				yi += xv.x * t.y + xv.y * t.x;
			} else {                                      // mac_real: imag of the tap ignored
				yr += xv.x * t.x;
				yi += xv.y * t.x;
			}
		}
		y[v * (size_t)y_len + i] = make_float2(yr, yi);
	}
}

extern "C" int trx_launch_convolve(const float *d_x, int x_len, const float *d_h, int h_len, int h_complex,
				   float *d_y, int y_len, int start, int len, size_t n_vec, hipStream_t stream)
{
	const size_t total = n_vec * (size_t)len;
	if (total == 0)
		return 0;
	const c32 *x = reinterpret_cast<const c32 *>(d_x);
	const size_t span_bytes = ((size_t)len + h_len - 1) * sizeof(c32);
	if (span_bytes <= 48 * 1024 && n_vec <= 0x7fffffffu && len >= 64) {  // one workgroup per vector, its window in LDS
		const c32 *hh = reinterpret_cast<const c32 *>(d_h);
		c32 *yy = reinterpret_cast<c32 *>(d_y);
		if (h_complex)
			hipLaunchKernelGGL(convolve_lds_kernel<true>, dim3((unsigned)n_vec), dim3(256), span_bytes, stream, x, x_len, hh, h_len, yy,
					   y_len, start, len);
		else
			hipLaunchKernelGGL(convolve_lds_kernel<false>, dim3((unsigned)n_vec), dim3(256), span_bytes, stream, x, x_len, hh, h_len, yy,
					   y_len, start, len);
		return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
	}
	size_t blocks = (total + 255) / 256;
	if (blocks > 256 * 8) blocks = 256 * 8;
	const c32 *h = reinterpret_cast<const c32 *>(d_h);
	c32 *y = reinterpret_cast<c32 *>(d_y);
	if (h_complex)
		hipLaunchKernelGGL(convolve_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, x, x_len, h, h_len, y,
				   y_len, start, len, n_vec);
	else
		hipLaunchKernelGGL(convolve_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, x_len, h, h_len, y,
				   y_len, start, len, n_vec);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// Channelizer(4, blockLen, 16)::rotate over a continuous wideband int16 stream.
//   path p takes wideband samples (M-1-p), (M-1-p)+M, ...                 (deinterleave, :37-48)
//   y_p[T] = sum_k xp[T-15+k] * sub_p[k]   with the previous block as history   (:86-94)
//   X_c[T] = 4-point forward DFT over p of y_p[T]                          (cxvec_fft, :96)
// Block boundaries of the reference are invisible in the maths (history carry == continuous stream,
// zero history before the first sample).  A workgroup takes CH_TILE consecutive output times; a thread computes
// CH_J = 4 consecutive ones for all 4 channels, so that the 16-tap windows of its outputs share their samples: 19 LDS
// reads per path for 4 outputs instead of 64 (the one-output-per-thread version of round 1 was bound by LDS
// bandwidth, not by HBM).  The staged samples are kept per path in a 4-phase layout -- xs[p][t & 3][t >> 2] -- so
// that lane l's sample t = 4 l + v sits at entry l + (v >> 2) of phase v & 3: consecutive lanes read consecutive
// 8-byte entries (bank-conflict-free) although each lane advances by 4 samples.  Sums run k = 0..15 per output,
// product then add, as convolve_base.c:41-54 does.
// ------------------------------------------------------------------------------------------------
#define CH_M 4
#define CH_H 16
#define CH_TPB 256
#define CH_J 4
#define CH_TILE (CH_TPB * CH_J)
#define CH_PHA 264                     // entries per phase array: >= (CH_TILE + 15 + 3) / 4 = 260; 2 * 264 = 16 (mod 64) dwords, so
                                       // the four phases of one loader pass fall on disjoint bank groups

typedef float ch_v2f __attribute__((ext_vector_type(2)));
template <int HI>
__device__ __forceinline__ ch_v2f ch_mul_tap(ch_v2f x, ch_v2f hpair)
{
	ch_v2f r;                          // x * (tap HI of the pair): one v_pk_mul_f32, the tap picked by op_sel
	if (HI)
		asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(hpair));
	else
		asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(hpair));
	return r;
}

// one complex sample from LDS as its own ds_read_b64 (a merged ds_read2_b64 occupies the LDS twice as long as two single
// reads, MI355X_MICROARCH.md); volatile only stops the merge
__device__ __forceinline__ ch_v2f ch_lds(const c32 *p)
{
	typedef const volatile ch_v2f __attribute__((address_space(3))) *lds_ptr;
	return *(lds_ptr)(p);
}

__global__ void __launch_bounds__(CH_TPB) __attribute__((amdgpu_waves_per_eu(4, 4)))     // 4 workgroups of 34 KB LDS per CU
channelize_kernel(const uint32_t *__restrict__ in, c32 *__restrict__ out, size_t n_total, size_t out_stride,
		  const trx_tables *__restrict__ tab, const uint4 *__restrict__ hist)
{
	// wideband time steps (T0-15) .. (T0+CH_TILE-1) as fp32, per path and phase: xs[p][t & 3][t >> 2], t = T - (T0-15)
	__shared__ __attribute__((aligned(16))) c32 xs[CH_M][4][CH_PHA];
	__shared__ __attribute__((aligned(16))) float taps[CH_M][CH_H];
	if (threadIdx.x < CH_M * CH_H)
		taps[threadIdx.x / CH_H][threadIdx.x % CH_H] = tab->chan_taps[threadIdx.x / CH_H][threadIdx.x % CH_H];
	const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
	auto stage = [&](int t, uint4 u) {         // time step t of the tile (0..14 = history) -> the four paths' fp32 samples
		const uint32_t w[4] = { u.x, u.y, u.z, u.w };
#pragma unroll
		for (int n = 0; n < CH_M; n++)         // path M-1-n <- wideband sample n of the time step (Channelizer.cpp:37-48)
			xs[CH_M - 1 - n][t & 3][t >> 2] = make_float2((float)(int16_t)(w[n] & 0xffffu), (float)(int16_t)(w[n] >> 16));
	};

	// A workgroup owns a contiguous run of tiles: the last 15 time steps of one tile are the history of the next, so every
	// tile costs exactly CH_J aligned 16-byte loads per thread, and those of tile k+1 are issued before tile k is
	// computed (register prefetch: the loads stay in flight during the arithmetic instead of in front of a barrier).
	const size_t n_tiles = (n_total + CH_TILE - 1) / CH_TILE;
	const size_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
	const size_t tile_lo = (size_t)blockIdx.x * per_wg;
	const size_t tile_hi = (tile_lo + per_wg < n_tiles) ? tile_lo + per_wg : n_tiles;
	uint4 pre[CH_J], carry = make_uint4(0u, 0u, 0u, 0u);
	auto prefetch = [&](size_t tile) {
#pragma unroll
		for (int i = 0; i < CH_J; i++) {
			const size_t ts = tile * CH_TILE + (size_t)i * CH_TPB + threadIdx.x;
			pre[i] = (ts < n_total) ? in4[ts] : make_uint4(0u, 0u, 0u, 0u);
		}
	};
	if (tile_lo < tile_hi) {
		prefetch(tile_lo);
		if (threadIdx.x >= CH_TPB - (CH_H - 1)) {                              // history of the run's first tile, from memory
			const long long ts = (long long)(tile_lo * CH_TILE) - CH_TPB + threadIdx.x;   // time steps T0-15 .. T0-1
			if (ts >= 0)
				carry = in4[ts];
			else if (hist)
				carry = hist[(CH_H - 1) + ts];                                 // carried history of a stream: time steps -15..-1
		}
	}
	for (size_t tile = tile_lo; tile < tile_hi; tile++) {
		const size_t T0 = tile * CH_TILE;
		__syncthreads();
		if (threadIdx.x >= CH_TPB - (CH_H - 1))
			stage(threadIdx.x - (CH_TPB - (CH_H - 1)), carry);                 // t = 0..14
#pragma unroll
		for (int i = 0; i < CH_J; i++)
			stage((CH_H - 1) + i * CH_TPB + threadIdx.x, pre[i]);
		carry = pre[CH_J - 1];                                                 // threads 241..255: the tile's last 15 time steps
		__syncthreads();
		if (tile + 1 < tile_hi)
			prefetch(tile + 1);
		const size_t T = T0 + (size_t)CH_J * threadIdx.x;                     // first of this thread's CH_J output times
		if (T < n_total) {
			c32 yp[CH_J][CH_M];
#pragma unroll
			for (int p = 0; p < CH_M; p++) {
				// samples v = 0 .. 18 of the thread's window: tap k of output j is sample j + k
				ch_v2f x[CH_J + CH_H - 1];
#pragma unroll
				for (int v = 0; v < CH_J + CH_H - 1; v++) {
					x[v] = ch_lds(&xs[p][v & 3][threadIdx.x + (v >> 2)]);
				}
				const float2 *g2 = reinterpret_cast<const float2 *>(&taps[p][0]);   // broadcast reads, a pair of taps each
				ch_v2f acc[CH_J];
#pragma unroll
				for (int j = 0; j < CH_J; j++)
					acc[j] = (ch_v2f){ 0.0f, 0.0f };
#pragma unroll
				for (int k = 0; k < CH_H; k++) {
					const float2 gq = g2[k >> 1];
					const ch_v2f gp = (ch_v2f){ gq.x, gq.y };
#pragma unroll
					for (int j = 0; j < CH_J; j++)
						acc[j] = acc[j] + ((k & 1) ? ch_mul_tap<1>(x[j + k], gp) : ch_mul_tap<0>(x[j + k], gp));
					if (k & 1)
						__builtin_amdgcn_sched_barrier(0);                     // keeps the products from piling up in registers
				}
#pragma unroll
				for (int j = 0; j < CH_J; j++) {
					asm volatile("" : "+v"(acc[j]));                           // the sums are finished here (not sunk into the store branches)
					yp[j][p] = make_float2(acc[j].x, acc[j].y);
				}
				__builtin_amdgcn_sched_barrier(0);                             // one path's window in registers at a time
			}
			// forward 4-point DFT per output time, radix-2 butterflies (exact +-1 / +-j twiddles)
			c32 o[CH_M][CH_J];
#pragma unroll
			for (int j = 0; j < CH_J; j++) {
				const c32 t1 = make_float2(yp[j][0].x + yp[j][2].x, yp[j][0].y + yp[j][2].y);
				const c32 t2 = make_float2(yp[j][0].x - yp[j][2].x, yp[j][0].y - yp[j][2].y);
				const c32 t3 = make_float2(yp[j][1].x + yp[j][3].x, yp[j][1].y + yp[j][3].y);
				const c32 t4 = make_float2(yp[j][1].x - yp[j][3].x, yp[j][1].y - yp[j][3].y);
				o[0][j] = make_float2(t1.x + t3.x, t1.y + t3.y);
				o[1][j] = make_float2(t2.x + t4.y, t2.y - t4.x);               // t2 - j*t4
				o[2][j] = make_float2(t1.x - t3.x, t1.y - t3.y);
				o[3][j] = make_float2(t2.x - t4.y, t2.y + t4.x);               // t2 + j*t4
			}
			// 32 contiguous bytes per channel and thread: two 16-byte stores when the row allows it
			const bool vec = (T + CH_J <= n_total) && ((out_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
			for (int c = 0; c < CH_M; c++) {
				c32 *dst = out + c * out_stride + T;
				if (vec) {
					reinterpret_cast<float4 *>(dst)[0] = make_float4(o[c][0].x, o[c][0].y, o[c][1].x, o[c][1].y);
					reinterpret_cast<float4 *>(dst)[1] = make_float4(o[c][2].x, o[c][2].y, o[c][3].x, o[c][3].y);
				} else {
#pragma unroll
					for (int j = 0; j < CH_J; j++)
						if (T + j < n_total)
							dst[j] = o[c][j];
				}
			}
		}
	}
}

// tail of a chunk -> history for the next one: the last 15 time steps (wideband) / samples (per channel)
__global__ void save_wide_hist_kernel(const uint4 *__restrict__ in4, size_t n_total, uint4 *__restrict__ hist)
{
	const int t = threadIdx.x;
	if (t < CH_H - 1) {
		const long long ts = (long long)n_total - (CH_H - 1) + t;
		const uint4 v = (ts >= 0) ? in4[ts] : hist[t + (int)n_total];       // chunk shorter than the history: shift
		__syncthreads();
		hist[t] = v;
	}
}

__global__ void save_chan_hist_kernel(const c32 *__restrict__ x, size_t n_in, size_t in_stride, c32 *__restrict__ hist)
{
	const int t = threadIdx.x, c = blockIdx.x;
	if (t < 15) {
		const long long s = (long long)n_in - 15 + t;
		const c32 v = (s >= 0) ? x[c * in_stride + s] : hist[c * 16 + t + (int)n_in];
		__syncthreads();
		hist[c * 16 + t] = v;
	}
}

extern "C" int trx_launch_channelize(const int16_t *d_in, float *d_out, size_t n_total, size_t out_stride,
				     const trx_tables *d_tab, void *d_hist_io, hipStream_t stream)
{
	if (n_total == 0)
		return 0;
	size_t blocks = (n_total + CH_TILE - 1) / CH_TILE;
	if (blocks > 256 * 4) blocks = 256 * 4;                              // 4 workgroups of 34 KB LDS per CU, each a run of tiles
	hipLaunchKernelGGL(channelize_kernel, dim3((unsigned)blocks), dim3(CH_TPB), 0, stream,
			   reinterpret_cast<const uint32_t *>(d_in), reinterpret_cast<c32 *>(d_out), n_total, out_stride, d_tab,
			   reinterpret_cast<const uint4 *>(d_hist_io));
	if (d_hist_io)
		hipLaunchKernelGGL(save_wide_hist_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const uint4 *>(d_in),
				   n_total, reinterpret_cast<uint4 *>(d_hist_io));
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// Resampler(p, q, 16)::rotate over a continuous stream per channel:
//   out[I] = sum_k in[n - 15 + k] * part[path][k],  n = (q*I)/p,  path = (q*I)%p     (Resampler.cpp:139-147,157-162)
// (per-block processing with history splice in the reference == the continuous formula, because
//  q*out_block == p*in_block; zero history before the first sample)
// Tiling: one workgroup handles TM periods = p*TM outputs from q*TM inputs (+15 of history) staged in LDS with
// coalesced loads; outputs are written coalesced; taps sit in LDS as [k][path] so that lanes (different
// paths) spread over the banks.  Index math is 32-bit inside a tile (the tile base is a multiple of the period).
// ------------------------------------------------------------------------------------------------
#define RS_TPB 256
#define RS_TILE_IN 3072                                                  // about q*TM input samples per tile

// Outputs o and o + p*m share their filter path ((q*o) % p), so a thread that walks o = t, t + S, t + 2S, ... with
// S = p*m (m = ceil(256 / p): S = 260 for 65/48, 65/96 and 52/75) keeps its 16 taps in registers for the whole kernel and
// reads only its 16 input samples from LDS per output; its input index advances by q*m per step.  The S - 256 output
// residues no thread owns (4 of 260) are swept afterwards with the taps read from LDS per output.
__global__ void __launch_bounds__(RS_TPB)
resample_kernel(const c32 *__restrict__ in, c32 *__restrict__ out, size_t n_in, size_t n_out, int p, int q, int tm, int m,
		size_t n_tiles, size_t in_stride, size_t out_stride, const float *__restrict__ parts,
		const c32 *__restrict__ hist)
{
	extern __shared__ __attribute__((aligned(16))) char rs_smem[];
	c32 *xs = reinterpret_cast<c32 *>(rs_smem);                          // [15 + q*tm]
	float *taps = reinterpret_cast<float *>(xs + 16 + q * tm);           // [16][p + 1]
	const int pst = p + 1;
	for (int i = threadIdx.x; i < p * 16; i += RS_TPB)
		taps[(i % 16) * pst + (i / 16)] = parts[i];
	const size_t chan = blockIdx.y;
	const c32 *x = in + chan * in_stride;
	c32 *y = out + chan * out_stride;
	const int tile_in = q * tm, tile_out = p * tm;
	const int S = p * m, iters = tm / m;                                 // tm is a multiple of m (launcher)
	const int t = threadIdx.x;
	const bool owner = t < S;                                            // (S >= 256 unless p > 256: then S = p and some threads idle)
	const unsigned qt = (unsigned)q * (unsigned)t;
	const int n_t = (int)(qt / (unsigned)p), path_t = (int)(qt % (unsigned)p);
	ch_v2f h2[8];                                                        // this thread's 16 taps, in pairs
#pragma unroll
	for (int k = 0; k < 8; k++)
		h2[k] = owner ? (ch_v2f){ parts[path_t * 16 + 2 * k], parts[path_t * 16 + 2 * k + 1] } : (ch_v2f){ 0.0f, 0.0f };
	const int nstep = q * m;
	// a workgroup owns a contiguous run of tiles and fetches tile k+1 into registers before it computes tile k
	constexpr int NPRE = (RS_TILE_IN + RS_TPB - 1) / RS_TPB;             // tile_in <= RS_TILE_IN
	const size_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
	const size_t tile_lo = (size_t)blockIdx.x * per_wg;
	const size_t tile_hi = (tile_lo + per_wg < n_tiles) ? tile_lo + per_wg : n_tiles;
	c32 pre[NPRE], pre_h = make_float2(0.0f, 0.0f);
	auto prefetch = [&](size_t tile) {
		const long long n0 = (long long)tile * tile_in;                    // first input sample of the tile
#pragma unroll
		for (int i = 0; i < NPRE; i++) {
			const int j = i * RS_TPB + t;
			const size_t sidx = (size_t)n0 + j;
			pre[i] = (j < tile_in && sidx < n_in) ? x[sidx] : make_float2(0.0f, 0.0f);
		}
		if (t < 15) {                                                      // the 15 samples in front of the tile
			const long long sidx = n0 - 15 + t;
			pre_h = make_float2(0.0f, 0.0f);
			if (sidx >= 0) { if ((size_t)sidx < n_in) pre_h = x[sidx]; }
			else if (hist) pre_h = hist[chan * 16 + 15 + sidx];                // carried history: samples -15..-1
		}
	};
	if (tile_lo < tile_hi)
		prefetch(tile_lo);
	for (size_t tile = tile_lo; tile < tile_hi; tile++) {
		__syncthreads();
		if (t < 15)
			xs[t] = pre_h;
#pragma unroll
		for (int i = 0; i < NPRE; i++) {
			const int j = i * RS_TPB + t;
			if (j < tile_in)
				xs[15 + j] = pre[i];
		}
		__syncthreads();
		if (tile + 1 < tile_hi)
			prefetch(tile + 1);
		const size_t o0 = tile * (size_t)tile_out;
		if (owner) {
			const c32 *xp = xs + n_t;                                      // xs[j] = in[n0 - 15 + j]
			c32 *yo = y + o0 + t;
			size_t o = o0 + t;
			for (int it = 0; it < iters && o < n_out; it++, o += S, xp += nstep, yo += S) {
				ch_v2f acc = { 0.0f, 0.0f };
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const ch_v2f xv = ch_lds(xp + k);
					acc = acc + ((k & 1) ? ch_mul_tap<1>(xv, h2[k >> 1]) : ch_mul_tap<0>(xv, h2[k >> 1]));   // product, then sum
				}
				*yo = make_float2(acc.x, acc.y);
			}
		}
		// residues RS_TPB .. S-1 of every step: (S - RS_TPB) * iters outputs, taps from LDS
		const int nres = S - RS_TPB;
		for (int idx = t; idx < nres * iters; idx += RS_TPB) {
			const int o = RS_TPB + idx % nres + S * (idx / nres);
			if (o0 + o >= n_out)
				continue;
			const unsigned qi = (unsigned)q * (unsigned)o;
			const int n = (int)(qi / (unsigned)p), path = (int)(qi % (unsigned)p);
			const c32 *xp = xs + n;
			float yr = 0.0f, yi = 0.0f;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const c32 xv = xp[k];
				const float h = taps[k * pst + path];
				yr += xv.x * h;
				yi += xv.y * h;
			}
			y[o0 + o] = make_float2(yr, yi);
		}
	}
}

extern "C" int trx_launch_resample(const float *d_in, float *d_out, size_t n_in, int p, int q, size_t n_chan,
				   size_t in_stride, size_t out_stride, const float *parts, void *d_hist_io, hipStream_t stream)
{
	const size_t n_out = n_in / q * p;
	if (n_chan * n_out == 0)
		return 0;
	const int m = (RS_TPB + p - 1) / p;                                  // outputs o and o + p*m share a filter path
	int tm = RS_TILE_IN / q / m * m;                                     // periods per tile: a multiple of m
	if (tm < m) tm = m;
	const size_t n_tiles = (n_out + (size_t)p * tm - 1) / ((size_t)p * tm);
	size_t gx = n_tiles;
	const size_t gmax = 1024 / n_chan > 0 ? 1024 / n_chan : 1;           // 4 workgroups (29 KB of LDS, 112 VGPRs) per CU over all channels,
	if (gx > gmax) gx = gmax;                                            // each walking a contiguous run of tiles
	const size_t lds = (size_t)(16 + q * tm) * sizeof(c32) + (size_t)16 * (p + 1) * sizeof(float);
	hipLaunchKernelGGL(resample_kernel, dim3((unsigned)gx, (unsigned)n_chan), dim3(RS_TPB), lds, stream,
			   reinterpret_cast<const c32 *>(d_in), reinterpret_cast<c32 *>(d_out), n_in, n_out, p, q, tm, m, n_tiles,
			   in_stride, out_stride, parts, reinterpret_cast<const c32 *>(d_hist_io));
	if (d_hist_io)
		hipLaunchKernelGGL(save_chan_hist_kernel, dim3((unsigned)n_chan), dim3(64), 0, stream,
				   reinterpret_cast<const c32 *>(d_in), n_in, in_stride, reinterpret_cast<c32 *>(d_hist_io));
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// Channelizer(4, ., 16)::rotate and Resampler(p, q, 16)::rotate of the four channels in ONE pass (round 3): the channel-rate
// streams never touch HBM.  The two kernels above move  16 B in + 32 B out  and  32 B in + 32 p/q B out  per wideband
// time step; fused it is 16 B in + 32 p/q B out -- 59 B instead of 123 B for 65/48.
// A workgroup owns a run of tiles of tm resampler periods = q*tm channel-rate time steps (960 for 65/48).  Per tile:
//   1. the wideband steps [T0 - 30, T0 + q*tm) are staged as fp32 in the channelizer's 4-phase layout (prefetched into
//      registers while the previous tile is computed; the 30 steps of overlap with the previous tile come from L2);
//   2. every thread computes 4 consecutive channel-rate times of all 4 channels -- the same 16-tap sums and the same
//      4-point DFT as channelize_kernel -- for the times [T0 - 15, T0 + q*tm): the resampler's 15 samples of history are
//      recomputed, not carried (the stream's very first tile takes them from the carried history instead);
//   3. behind a barrier the channel samples go to LDS OVER the staged wideband samples (34 KB per workgroup, four
//      workgroups per CU as before);
//   4. every thread resamples the outputs t, t + S, ... (S = p*m: same filter path, taps in registers) of all four
//      channels, exactly as resample_kernel does, and the residues S - 256 .. S - 1 are swept afterwards.
// Arithmetic and operand order are those of the two kernels: the results are bit-identical to running them in sequence
// (tests/test_gpu_aux_kernels.py).  The last tile's workgroup leaves the final 15 channel samples in hist_out.
// ------------------------------------------------------------------------------------------------
#define FE_CS 1056                      // entries per channel in the aliased LDS array (>= q*tm + 15, <= 4 * CH_PHA)

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, not for its global loads and stores.
// __syncthreads() also drains vmcnt -- here that would wait, at every phase change, for the previous tile's output stores
// to be acknowledged by HBM and for the next tile's prefetch loads to land, which is most of what the two-kernel form spends.
__device__ __forceinline__ void fe_lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ void __launch_bounds__(CH_TPB) __attribute__((amdgpu_waves_per_eu(4, 4)))
frontend_fused_kernel(const uint4 *__restrict__ in4, size_t n_total, c32 *__restrict__ out, size_t n_out, size_t out_stride,
		      int p, int q, int tm, int m, size_t n_tiles, const float *__restrict__ parts,
		      const trx_tables *__restrict__ tab, const uint4 *__restrict__ wide_hist,
		      const c32 *__restrict__ chan_hist_in, c32 *__restrict__ chan_hist_out)
{
	__shared__ __attribute__((aligned(16))) c32 xs[CH_M][4][CH_PHA];     // wideband staging, then cs[CH_M][FE_CS]
	__shared__ __attribute__((aligned(16))) float taps[CH_M][CH_H];
	extern __shared__ __attribute__((aligned(16))) char fe_smem[];        // resampler taps [16][p + 1] (residue sweep)
	float *rtaps = reinterpret_cast<float *>(fe_smem);
	c32 *const cs = &xs[0][0][0];
	static_assert(CH_M * FE_CS <= CH_M * 4 * CH_PHA, "the channel samples alias the wideband staging");
	const int t = threadIdx.x;
	const int pst = p + 1;
	if (t < CH_M * CH_H)
		taps[t / CH_H][t % CH_H] = tab->chan_taps[t / CH_H][t % CH_H];
	for (int i = t; i < p * 16; i += CH_TPB)
		rtaps[(i % 16) * pst + (i / 16)] = parts[i];
	const int tile_in = q * tm, tile_out = p * tm;
	const int n_stage = tile_in + 30;                                    // wideband steps staged per tile (<= 4 * CH_TPB)
	const int n_cs = tile_in + 15;                                       // channel-rate times computed per tile
	// Resampling as in resample_kernel: thread t owns the outputs t, t + S, ... (S = p*m: the same filter path at every step,
	// taps in registers).  (A variant with output PAIRS per thread -- 17 LDS reads for two outputs instead of 32, the second
	// output's taps shifted inside 17 entries -- read a third less from LDS and was slower: smaller tiles, more of the
	// per-tile latency chain below.  The kernel is bound by that chain, not by LDS or arithmetic; profiles/r03_ab_runs.txt.)
	const int S = p * m, iters = tm / m;
	const bool owner = t < S;
	const unsigned qt = (unsigned)q * (unsigned)t;
	const int n_t0 = (int)(qt / (unsigned)p), path_t0 = (int)(qt % (unsigned)p);
	const int nstep = q * m;

	// The residues S - 256 .. S - 1 of every resampler step (all channels) are swept item by item.  Which output, which filter
	// path and which input sample an item is depends on the item alone, not on the tile: worked out once, here, and parked in
	// LDS as two packed words -- per tile the four integer divisions (~ 160 instructions on the waves that hold items, which
	// the other waves of the workgroup then wait for at the next barrier) become one 8-byte read.
	// (first sample, filter path) of this thread's output positions: a function of the thread alone, but kept in two registers
	// across the tile loop it was what the allocator spilled at 128 -- and a scratch reload in front of the resampler loop waits
	// for vmcnt(0), i.e. for the next tile's prefetch that has just been issued.  Parked in LDS, one ds_read_b32 per tile.
	int *const thr_item = reinterpret_cast<int *>(rtaps + 16 * pst);
	thr_item[t] = (n_t0 << 16) | path_t0;
	const int n_res = (S - CH_TPB) * iters * CH_M;
	int2 *const res_item = reinterpret_cast<int2 *>(rtaps + 16 * pst + CH_TPB);
	for (int idx = t; idx < n_res; idx += CH_TPB) {
		const int nres0 = S - CH_TPB;
		const int c = idx / (nres0 * iters), r = idx % (nres0 * iters);
		const int oi = CH_TPB + r % nres0 + S * (r / nres0);
		const unsigned qi = (unsigned)q * (unsigned)oi;
		res_item[idx] = make_int2((c << 16) | (int)(qi / (unsigned)p), (oi << 16) | (int)(qi % (unsigned)p));
	}
	const size_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
	const size_t tile_lo = (size_t)blockIdx.x * per_wg;
	const size_t tile_hi = (tile_lo + per_wg < n_tiles) ? tile_lo + per_wg : n_tiles;
	uint4 pre[4];
	auto prefetch = [&](size_t tile) {
		// staged step j (0 <= j < n_stage) is wideband step t0 + j.  Everything that depends on the tile is wave-uniform: the
		// range [jlo, jhi) of steps inside the stream and a base pointer; a thread adds its 32-bit j.  (Round 5: the 64-bit
		// per-thread step numbers this replaces were spilled to scratch, and every reload waited for vmcnt(0) -- for the
		// previous tile's output stores -- in the middle of the tile loop.)
		const long long t0 = (long long)tile * tile_in - 30;               // first staged wideband step
		const long long lo = t0 < 0 ? -t0 : 0, hi = (long long)n_total - t0;
		const int jlo = (int)(lo < n_stage ? lo : n_stage), jhi = (int)(hi < 0 ? 0 : (hi < n_stage ? hi : n_stage));
		const uint4 *const base = in4 + t0;                                 // (dereferenced for jlo <= j < jhi only)
		const int hoff = (int)((CH_H - 1) + t0);                            // tile 0: steps -15 .. -1 come from the carried history
		if (jlo == 0 && jhi == n_stage) {
			// a tile inside the stream (all but the first and the last): four unconditional loads.  A branch of its own, and
			// wave-uniform -- with the edge cases as the other arm of a per-lane if, both arms ran one after the other in every
			// wave, both wrote pre[i], and the compiler put an s_waitcnt vmcnt(0) between them: loads three and four of every
			// tile waited for loads one and two to come back from HBM.
			unsigned tt = (unsigned)t;                                       // opaque per tile: the four clamped offsets are three instructions
			asm volatile("" : "+v"(tt));                                    // each to recompute, and 8 registers (spilled) to keep across tiles
			const unsigned jmax = (unsigned)n_stage - 1u;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const unsigned j = (unsigned)(i * CH_TPB) + tt;
				pre[i] = base[j < jmax ? j : jmax];                            // (j >= n_stage is never staged)
			}
			return;
		}
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int j = i * CH_TPB + t;
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if (j >= jlo && j < jhi)
				v = base[j];
			else if (j < jlo && hoff + j >= 0 && wide_hist)
				v = wide_hist[hoff + j];
			pre[i] = v;
		}
	};
	if (tile_lo < tile_hi)
		prefetch(tile_lo);
	for (size_t tile = tile_lo; tile < tile_hi; tile++) {
		fe_lds_barrier();                                                   // (the previous tile's channel samples are done with)
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int j = i * CH_TPB + t;
			if (j < n_stage) {
				const uint32_t w[4] = { pre[i].x, pre[i].y, pre[i].z, pre[i].w };
#pragma unroll
				for (int n = 0; n < CH_M; n++)                                 // path M-1-n <- wideband sample n of the time step
					xs[CH_M - 1 - n][j & 3][j >> 2] = make_float2((float)(int16_t)(w[n] & 0xffffu), (float)(int16_t)(w[n] >> 16));
			}
		}
		fe_lds_barrier();

		// ---- channelizer: channel-rate times u = 4t .. 4t+3 of the tile (time T0 - 15 + u); tap k of output u is staged step u + k
		c32 o[CH_M][CH_J];
		const bool active = CH_J * t < n_cs;
		if (active) {
			c32 yp[CH_J][CH_M];
#pragma unroll
			for (int pp = 0; pp < CH_M; pp++) {
				// The window is requested LAST sample first: the first product needs x[0], the newest request, and its s_waitcnt
				// covers the whole window (LDS returns in order) -- one wait per path where the ascending order had one per tap.
				ch_v2f x[CH_J + CH_H - 1];
#pragma unroll
				for (int v = CH_J + CH_H - 2; v >= 0; v--)
					x[v] = ch_lds(&xs[pp][v & 3][t + (v >> 2)]);
				// (round 5 measured these 16 wave-uniform taps as scalar operands of the packed multiplies, fetched per path through
				// the scalar cache instead of LDS: 1.30 ms against 1.04 -- the s_load and its lgkmcnt wait sit in front of every
				// path's sums; profiles/r05_ab_runs.txt)
				// A pair of taps is one broadcast 8-byte read, requested one pair AHEAD of its use (the compiler had sunk each read
				// to its first use: request, s_waitcnt lgkmcnt(0), multiply -- a full LDS round trip in front of every tap pair).
				// Products of tap k alternate with the sums of tap k - 1: a dependent packed pair costs a pad, and a packed
				// instruction in between does not count as one (see the resampler loop below).
				typedef const volatile ch_v2f __attribute__((address_space(3))) *lds_tap;
				const lds_tap g2 = (lds_tap)(&taps[pp][0]);
				ch_v2f acc[CH_J], pr[CH_J];
#pragma unroll
				for (int j = 0; j < CH_J; j++)
					acc[j] = pr[j] = (ch_v2f){ 0.0f, 0.0f };
				ch_v2f gcur = g2[0];
#pragma unroll
				for (int kp = 0; kp < CH_H / 2; kp++) {
					ch_v2f gnext = gcur;
					if (kp + 1 < CH_H / 2)
						gnext = g2[kp + 1];
#pragma unroll
					for (int kk = 0; kk < 2; kk++) {
						const int k = 2 * kp + kk;
#pragma unroll
						for (int j = 0; j < CH_J; j++) {
							const ch_v2f pn = kk ? ch_mul_tap<1>(x[j + k], gcur) : ch_mul_tap<0>(x[j + k], gcur);
							if (k > 0)
								acc[j] = acc[j] + pr[j];                               // product, then sum: tap k - 1
							pr[j] = pn;
							__builtin_amdgcn_sched_barrier(0);
						}
					}
					gcur = gnext;
				}
#pragma unroll
				for (int j = 0; j < CH_J; j++)
					acc[j] = acc[j] + pr[j];                                           // tap 15
#pragma unroll
				for (int j = 0; j < CH_J; j++) {
					asm volatile("" : "+v"(acc[j]));
					yp[j][pp] = make_float2(acc[j].x, acc[j].y);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
#pragma unroll
			for (int j = 0; j < CH_J; j++) {                                   // forward 4-point DFT (exact +-1 / +-j twiddles)
				const c32 t1 = make_float2(yp[j][0].x + yp[j][2].x, yp[j][0].y + yp[j][2].y);
				const c32 t2 = make_float2(yp[j][0].x - yp[j][2].x, yp[j][0].y - yp[j][2].y);
				const c32 t3 = make_float2(yp[j][1].x + yp[j][3].x, yp[j][1].y + yp[j][3].y);
				const c32 t4 = make_float2(yp[j][1].x - yp[j][3].x, yp[j][1].y - yp[j][3].y);
				o[0][j] = make_float2(t1.x + t3.x, t1.y + t3.y);
				o[1][j] = make_float2(t2.x + t4.y, t2.y - t4.x);
				o[2][j] = make_float2(t1.x - t3.x, t1.y - t3.y);
				o[3][j] = make_float2(t2.x - t4.y, t2.y + t4.x);
			}
		}
		if (tile + 1 < tile_hi)                                            // (here, not before the filters: 16 registers they need)
			prefetch(tile + 1);
		fe_lds_barrier();                                                   // every window has been read: the staging area is free
		if (active) {
#pragma unroll
			for (int c = 0; c < CH_M; c++) {
				float4 *dst = reinterpret_cast<float4 *>(cs + c * FE_CS + CH_J * t);
				dst[0] = make_float4(o[c][0].x, o[c][0].y, o[c][1].x, o[c][1].y);
				dst[1] = make_float4(o[c][2].x, o[c][2].y, o[c][3].x, o[c][3].y);
			}
		}
		if (tile == 0) {                                                   // the stream's first tile: times -15 .. -1 are the carried history
			fe_lds_barrier();
			if (t < CH_M * 15)
				cs[(t / 15) * FE_CS + (t % 15)] = chan_hist_in ? chan_hist_in[(t / 15) * 16 + (t % 15)] : make_float2(0.0f, 0.0f);
		}
		fe_lds_barrier();

		// ---- resampler: cs[c][j] = channel c at time T0 - 15 + j
		const size_t o0 = tile * (size_t)tile_out;
		if (owner) {
			// this thread's 16 taps (path (q t) mod p), re-read from LDS per tile: 16 registers the channelizer above needs more
			// than this loop does.  Two CHANNELS of an output position per pass: the same taps and offsets, 32 LDS reads in
			// flight and two independent sum chains instead of one (the waves of this kernel wait two thirds of their time:
			// profiles/r04_ab_runs.txt); each output's sum still runs k = 0..15, product then add.
			typedef const volatile int __attribute__((address_space(3))) *lds_int;
			int tl = t;                                                     // (opaque: the address is one instruction to form per tile -- hoisted
			asm volatile("" : "+v"(tl));                                    // out of the tile loop it was the next value to be spilled)
			const int ti = *(lds_int)(thr_item + tl);
			const int n_t = ti >> 16, path_t = ti & 0xffff;
			ch_v2f h2[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
				h2[k] = (ch_v2f){ rtaps[(2 * k) * pst + path_t], rtaps[(2 * k + 1) * pst + path_t] };
			{
				// all four channels of an output position per pass: four independent sum chains over the same taps and offsets,
				// 16 LDS reads in flight per block of four taps
				const c32 *xa = cs + n_t;
				c32 *const ob = out + o0;                                   // wave-uniform base; the thread adds a 32-bit offset
				unsigned yo = (unsigned)t;                                  // (a per-thread 64-bit pointer here was spilled and reloaded --
				size_t oo = o0 + t;                                         // behind a vmcnt(0) wait -- once per tile)
				for (int it = 0; it < iters && oo < n_out; it++, oo += S, xa += nstep, yo += (unsigned)S) {
					ch_v2f acc[CH_M], pr[CH_M];
#pragma unroll
					for (int c = 0; c < CH_M; c++)
						acc[c] = pr[c] = (ch_v2f){ 0.0f, 0.0f };
#pragma unroll
					for (int k0 = 0; k0 < 16; k0 += 4) {
						ch_v2f x[CH_M][4];
#pragma unroll
						for (int k = 0; k < 4; k++)
#pragma unroll
							for (int c = 0; c < CH_M; c++)
								x[c][k] = ch_lds(xa + c * FE_CS + k0 + k);
						// A tap at a time, software-pipelined: the four channels' products of tap k (the LAST-read channel first -- its
						// s_waitcnt covers the other three, LDS returns in order) alternate with the four sums of tap k - 1.  Written as
						// product + sum per channel the compiler ran all sixteen through one product register: wait, multiply, s_nop
						// (the pad of a dependent packed pair), add -- four issue slots per tap and channel where two do the
						// arithmetic; a packed instruction between the two does not count as their pad, so products and sums of the
						// SAME tap in two groups of four still paid one s_nop per tap (round 5).
#pragma unroll
						for (int k = 0; k < 4; k++) {
							const int kk = k0 + k;
#pragma unroll
							for (int c = CH_M - 1; c >= 0; c--) {
								const ch_v2f pn = (kk & 1) ? ch_mul_tap<1>(x[c][k], h2[kk >> 1]) : ch_mul_tap<0>(x[c][k], h2[kk >> 1]);
								if (kk > 0)
									acc[c] = acc[c] + pr[c];                           // product, then sum: tap kk - 1
								pr[c] = pn;
								__builtin_amdgcn_sched_barrier(0);
							}
						}
					}
#pragma unroll
					for (int c = CH_M - 1; c >= 0; c--)
						acc[c] = acc[c] + pr[c];                                       // tap 15
#pragma unroll
					for (int c = 0; c < CH_M; c++)
						(ob + (size_t)c * out_stride)[yo] = make_float2(acc[c].x, acc[c].y);
				}
			}
		}
		int tr = t;                                                         // (opaque, as above: the item's address is formed per tile)
		asm volatile("" : "+v"(tr));
		for (int idx = tr; idx < n_res; idx += CH_TPB) {                   // residues of every step, all channels
			const int2 e = res_item[idx];
			const int c = e.x >> 16, n = e.x & 0xffff, oi = e.y >> 16, path = e.y & 0xffff;
			if (o0 + oi >= n_out)
				continue;
			const c32 *xp = cs + c * FE_CS + n;
			ch_v2f acc = { 0.0f, 0.0f };                                    // product, then sum, k ascending, on both components at once:
#pragma unroll
			for (int k = 0; k < 16; k++) {                                  // the two roundings per tap of yr += x.x * h, yi += x.y * h
				const float h = rtaps[k * pst + path];
				acc = acc + ch_lds(xp + k) * (ch_v2f){ h, h };
			}
			out[c * out_stride + o0 + oi] = make_float2(acc.x, acc.y);
		}
		if (chan_hist_out && tile + 1 == n_tiles && t < CH_M * 15) {        // the call's last 15 channel samples
			const long long j = (long long)n_total - (long long)tile * tile_in + (t % 15);   // time n_total - 15 + i -> cs index
			chan_hist_out[(t / 15) * 16 + (t % 15)] = cs[(t / 15) * FE_CS + j];
		}
	}
}

// fused front end; returns 1 when the geometry does not fit (the caller then runs the two kernels), 0 / TRXHIP_EIO otherwise
extern "C" int trx_launch_frontend_fused(const int16_t *d_wide, float *d_out, size_t n_total, int p, int q, size_t out_stride,
					 const float *parts, const trx_tables *d_tab, void *d_wide_hist_io, const void *d_chan_hist_in,
					 void *d_chan_hist_out, hipStream_t stream)
{
	const int m = (CH_TPB + p - 1) / p;                                  // outputs o and o + p*m share a filter path
	const int tm = (4 * CH_TPB - 30) / q / m * m;                         // periods per tile: staging fits 4 loads per thread
	const size_t n_out = n_total / q * p;
	if (tm < m || p * m < CH_TPB || p * m > 2 * CH_TPB || q * tm + 15 > FE_CS || (n_total % (size_t)q) != 0 || n_total < 30 || n_out == 0)
		return 1;
	const size_t n_tiles = (n_out + (size_t)p * tm - 1) / ((size_t)p * tm);
	size_t gx = n_tiles < 1024 ? n_tiles : 1024;                         // 4 workgroups of 34 KB LDS per CU, each a run of tiles
	const size_t n_res = (size_t)(p * m - CH_TPB) * (tm / m) * CH_M;    // residue items per tile (80 at 65 / 48)
	if (n_res > 1024)
		return 1;
	const size_t lds = (size_t)16 * (p + 1) * sizeof(float) + CH_TPB * sizeof(int) + n_res * sizeof(int2);   // resampler taps + the threads' items + the residue items
	hipLaunchKernelGGL(frontend_fused_kernel, dim3((unsigned)gx), dim3(CH_TPB), lds, stream, reinterpret_cast<const uint4 *>(d_wide),
			   n_total, reinterpret_cast<c32 *>(d_out), n_out, out_stride, p, q, tm, m, n_tiles, parts, d_tab,
			   reinterpret_cast<const uint4 *>(d_wide_hist_io), reinterpret_cast<const c32 *>(d_chan_hist_in),
			   reinterpret_cast<c32 *>(d_chan_hist_out));
	if (d_wide_hist_io)
		hipLaunchKernelGGL(save_wide_hist_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const uint4 *>(d_wide),
				   n_total, reinterpret_cast<uint4 *>(d_wide_hist_io));
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// TRXD payload packing (proto_trxd.c:36-66): one 156-byte record per burst, one wave per 4 bursts
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_trxd_kernel(const trxhip_burst_result *__restrict__ res, const float *__restrict__ soft, int soft_stride,
		 uint8_t *__restrict__ pkt, size_t n_bursts, float rssi_offset)
{
	const size_t total = n_bursts * 39;                                  // 39 dwords per record
	uint32_t *out = reinterpret_cast<uint32_t *>(pkt);
	for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
		const size_t b = o / 39;
		const int wd = (int)(o - b * 39);
		const trxhip_burst_result r = res[b];
		uint32_t word;
		if (wd == 0) {
			const int toa_int = (int)((double)r.toa * 256.0 + 0.5);        // trxd_fill_v0_specific
			double rssi = (double)r.rssi + (double)rssi_offset;
			uint32_t rssi_u8 = (rssi >= 255.0 || rssi != rssi) ? 255u : (rssi <= 0.0 ? 0u : (uint32_t)rssi);
			const int ci_cb = (int16_t)((double)(r.ci * 10) + 0.5);        // trxd_fill_v1_specific
			word = ((uint32_t)(toa_int >> 8) & 0xffu) | (((uint32_t)toa_int & 0xffu) << 8) | (rssi_u8 << 16) |
			       ((((uint32_t)ci_cb >> 8) & 0xffu) << 24);
		} else if (wd == 1) {
			const int ci_cb = (int16_t)((double)(r.ci * 10) + 0.5);
			word = ((uint32_t)ci_cb & 0xffu) | ((uint32_t)r.tsc << 8) | ((uint32_t)r.idle << 16) |
			       ((uint32_t)r.nbits_div4 << 24);
		} else {
			word = 0;
			const float *s = soft + b * (size_t)soft_stride + (wd - 2) * 4;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t u = r.idle ? 0u : (uint32_t)(uint8_t)round((double)s[k] * 255.0);   // normalized255
				word |= u << (8 * k);
			}
		}
		out[o] = word;
	}
}

extern "C" int trx_launch_pack_trxd(const trxhip_burst_result *d_results, const float *d_soft, int soft_stride,
				    uint8_t *d_pkt, size_t n_bursts, float rssi_offset, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	size_t blocks = (n_bursts * 39 + 255) / 256;
	if (blocks > 256 * 8) blocks = 256 * 8;
	hipLaunchKernelGGL(pack_trxd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_results, d_soft, soft_stride,
			   d_pkt, n_bursts, rssi_offset);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// TRXD v0 / v1 uplink burst indications in wire format (proto_trxd.c:28-117, proto_trxd.h:56-106): the datagram of
// burst b at pkt + b * pkt_stride, its length in pkt_len[b].  One thread per output dword (coalesced stores); each
// thread rebuilds the few header fields it needs from the 32-byte result record (L1/L2 hits) -- the kernel moves
// 32 + 592 B in and <= 160 B out per burst and is a small fraction of the detect/demod launch in front of it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t trxd_byte(int p, int hdr_len, int nbits, bool v1, const uint8_t *hdr, const float *s)
{
	if (p < hdr_len)
		return hdr[p];
	const int k = p - hdr_len;
	if (k < nbits)
		return (uint32_t)(uint8_t)round((double)s[k] * 255.0);      // trxd_fill_burst_normalized255(), :62-66
	return 0u;                                                           // v0's two trailing bytes (:83-87), padding
}

__global__ void __launch_bounds__(256)
pack_trxd_wire_kernel(const trxhip_burst_result *__restrict__ res, const trxhip_burst_params *__restrict__ prm,
		      const float *__restrict__ soft, int soft_stride, const trxhip_trxd_meta *__restrict__ meta,
		      uint8_t *__restrict__ pkt, int pkt_stride, uint16_t *__restrict__ pkt_len, size_t n_bursts, float rssi_offset,
		      trxhip_burst_result *__restrict__ res_copy)
{
	const int wpb = pkt_stride >> 2;                                     // dwords per burst
	const size_t total = n_bursts * (size_t)wpb;
	uint32_t *out = reinterpret_cast<uint32_t *>(pkt);
	for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
		const size_t b = o / wpb;
		const int wd = (int)(o - b * wpb);
		const trxhip_burst_result r = res[b];
		const trxhip_trxd_meta m = meta[b];
		const bool v1 = m.version != 0;
		const bool off = prm[b].type == TRXHIP_OFF;                      // -ENOENT: nothing is sent (Transceiver.cpp:704-707)
		const bool idle = r.idle != 0;
		int nbits = idle ? 0 : 4 * (int)r.nbits_div4;
		const int hdr_len = v1 ? TRXHIP_TRXD_V1_HDR : TRXHIP_TRXD_V0_HDR;
		int len = v1 ? hdr_len + nbits : hdr_len + nbits + 2;            // :76, :96-99
		if (off || (!v1 && idle))                                        // v0 drops idle indications (:71-73)
			len = 0;
		if (len > pkt_stride) {
			len = pkt_stride;
			nbits = nbits < pkt_stride - hdr_len ? nbits : pkt_stride - hdr_len;
		}
		if (nbits > soft_stride) nbits = soft_stride;

		uint8_t hdr[TRXHIP_TRXD_V1_HDR];
		hdr[0] = (uint8_t)(((v1 ? 1u : 0u) << 4) | (m.tn & 7u));         // trxd_fill_common(): version:4 | reserved:1 | tn:3
		hdr[1] = (uint8_t)(m.fn >> 24); hdr[2] = (uint8_t)(m.fn >> 16); hdr[3] = (uint8_t)(m.fn >> 8); hdr[4] = (uint8_t)m.fn;
		const double rssi = (double)r.rssi + (double)rssi_offset;        // bi->rssi (Transceiver.cpp:751)
		hdr[5] = (rssi >= 255.0) ? 255u : (rssi > 0.0 ? (uint8_t)rssi : 0u);   // v0->rssi = bi->rssi (NaN -> 0)
		const int toa_int = idle ? 0 : (int)((double)r.toa * 256.0 + 0.5);     // trxd_fill_v0_specific(), :36-45
		hdr[6] = (uint8_t)((uint32_t)toa_int >> 8); hdr[7] = (uint8_t)toa_int;
		const bool psk = !idle && r.nbits_div4 == 111;                   // bi->modulation (Transceiver.cpp:794-800)
		const uint32_t mod = psk ? (4u | (m.tss & 1u)) : (m.tss & 3u);   // TRXD_MODULATION_8PSK / _GMSK
		hdr[8] = (uint8_t)(((idle ? 1u : 0u) << 7) | (mod << 3) | (idle ? 0u : (r.tsc & 7u)));   // tsc:3 | modulation:4 | idle:1
		const int ci_cb = idle ? 0 : (int16_t)((double)(r.ci * 10) + 0.5);    // trxd_fill_v1_specific(), :47-60
		hdr[9] = (uint8_t)((uint32_t)ci_cb >> 8); hdr[10] = (uint8_t)ci_cb;

		const float *s = soft + b * (size_t)soft_stride;
		uint32_t word = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int p = 4 * wd + k;
			if (p < len)
				word |= trxd_byte(p, hdr_len, nbits, v1, hdr, s) << (8 * k);
		}
		out[o] = word;
		if (wd == 0) {
			pkt_len[b] = (uint16_t)len;
			if (res_copy)                                                // (host pipe: the record goes out with the datagram)
				res_copy[b] = r;
		}
	}
}

// Round 3: 16 datagram bytes per thread for rows that are a multiple of 16 bytes (the 160-byte rows of the host pipe).
// The per-burst fields (result record, meta word, type) are fetched by 10 threads per burst instead of 40, a thread behind
// the header and in front of the row's end reads its sixteen soft bits as four 4-byte-aligned 16-byte loads and quantises
// them with one round-to-nearest-even each (x in [0, 1]: x * 255 is exact in double and its only tie, 127.5 at x = 0.5,
// goes to 128 under round() and under rint() alike; anything outside [0, 1] takes the generic expression), and writes one
// 16-byte store.  Header and tail threads take the byte-by-byte form above.  (One WAVE per burst was tried and measured
// 0.65 ms against 0.41: the per-burst loads become exposed latency; a thread per dword with the header skipped: 0.48.)
__device__ __forceinline__ uint32_t trxd_q255(float x)
{
	return (x >= 0.0f && x <= 1.0f) ? (uint32_t)__builtin_rint((double)x * 255.0) : (uint32_t)(uint8_t)round((double)x * 255.0);
}

__global__ void __launch_bounds__(256)
pack_trxd_wire16_kernel(const trxhip_burst_result *__restrict__ res, const trxhip_burst_params *__restrict__ prm,
			const float *__restrict__ soft, int soft_stride, const trxhip_trxd_meta *__restrict__ meta,
			uint8_t *__restrict__ pkt, int pkt_stride, uint16_t *__restrict__ pkt_len, unsigned n_bursts, float rssi_offset,
			trxhip_burst_result *__restrict__ res_copy)
{
	const unsigned cpb = (unsigned)pkt_stride >> 4;                      // 16-byte chunks per burst
	const unsigned total = n_bursts * cpb;                               // (the launcher keeps this below 2^32)
	uint4 *out = reinterpret_cast<uint4 *>(pkt);
	for (unsigned o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
		const unsigned b = o / cpb;
		const int p0 = 16 * (int)(o - b * cpb);                          // first datagram byte of this thread
		const uint32_t rlast = reinterpret_cast<const uint32_t *>(res + b)[7];   // tsc | clip << 8 | idle << 16 | nbits / 4 << 24
		const trxhip_trxd_meta m = meta[b];
		const bool v1 = m.version != 0;
		const bool off = prm[b].type == TRXHIP_OFF;                      // -ENOENT: nothing is sent (Transceiver.cpp:704-707)
		const bool idle = ((rlast >> 16) & 0xffu) != 0;
		int nbits = idle ? 0 : 4 * (int)(rlast >> 24);
		const int hdr_len = v1 ? TRXHIP_TRXD_V1_HDR : TRXHIP_TRXD_V0_HDR;
		int len = v1 ? hdr_len + nbits : hdr_len + nbits + 2;            // :76, :96-99
		if (off || (!v1 && idle))                                        // v0 drops idle indications (:71-73)
			len = 0;
		if (len > pkt_stride) {
			len = pkt_stride;
			nbits = nbits < pkt_stride - hdr_len ? nbits : pkt_stride - hdr_len;
		}
		if (nbits > soft_stride) nbits = soft_stride;
		const float *s = soft + b * (size_t)soft_stride;
		const int q0 = p0 - hdr_len;                                     // first soft bit of this thread
		uint32_t w[4] = {0u, 0u, 0u, 0u};
		if (q0 >= 0 && q0 + 16 <= nbits && p0 + 16 <= len) {             // sixteen soft bits, nothing else
#pragma unroll
			for (int j = 0; j < 4; j++) {
				float4 v;
				__builtin_memcpy(&v, s + q0 + 4 * j, sizeof(v));         // 4-byte aligned: one global_load_dwordx4
				w[j] = trxd_q255(v.x) | (trxd_q255(v.y) << 8) | (trxd_q255(v.z) << 16) | (trxd_q255(v.w) << 24);
			}
		} else if (p0 < len) {
			uint8_t hdr[TRXHIP_TRXD_V1_HDR];
#pragma unroll
			for (int k = 0; k < TRXHIP_TRXD_V1_HDR; k++) hdr[k] = 0;
			if (p0 == 0) {                                               // the header lives in the first chunk (hdr_len <= 11)
				const trxhip_burst_result r = res[b];
				if (res_copy)                                                // (host pipe: the record goes out with the datagram)
					res_copy[b] = r;
				hdr[0] = (uint8_t)(((v1 ? 1u : 0u) << 4) | (m.tn & 7u));     // trxd_fill_common(): version:4 | reserved:1 | tn:3
				hdr[1] = (uint8_t)(m.fn >> 24); hdr[2] = (uint8_t)(m.fn >> 16); hdr[3] = (uint8_t)(m.fn >> 8); hdr[4] = (uint8_t)m.fn;
				const double rssi = (double)r.rssi + (double)rssi_offset;    // bi->rssi (Transceiver.cpp:751)
				hdr[5] = (rssi >= 255.0) ? 255u : (rssi > 0.0 ? (uint8_t)rssi : 0u);   // v0->rssi = bi->rssi (NaN -> 0)
				const int toa_int = idle ? 0 : (int)((double)r.toa * 256.0 + 0.5);     // trxd_fill_v0_specific(), :36-45
				hdr[6] = (uint8_t)((uint32_t)toa_int >> 8); hdr[7] = (uint8_t)toa_int;
				const bool psk = !idle && r.nbits_div4 == 111;               // bi->modulation (Transceiver.cpp:794-800)
				const uint32_t mod = psk ? (4u | (m.tss & 1u)) : (m.tss & 3u);   // TRXD_MODULATION_8PSK / _GMSK
				hdr[8] = (uint8_t)(((idle ? 1u : 0u) << 7) | (mod << 3) | (idle ? 0u : (r.tsc & 7u)));   // tsc:3 | modulation:4 | idle:1
				const int ci_cb = idle ? 0 : (int16_t)((double)(r.ci * 10) + 0.5);    // trxd_fill_v1_specific(), :47-60
				hdr[9] = (uint8_t)((uint32_t)ci_cb >> 8); hdr[10] = (uint8_t)ci_cb;
			}
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const int p = p0 + k;
				if (p < len)
					w[k >> 2] |= trxd_byte(p, hdr_len, nbits, v1, hdr, s) << (8 * (k & 3));
			}
		}
		out[o] = make_uint4(w[0], w[1], w[2], w[3]);
		if (p0 == 0) {
			pkt_len[b] = (uint16_t)len;
			if (res_copy && !(p0 < len))                                 // (nothing to send: the header branch did not run)
				res_copy[b] = res[b];
		}
	}
}

extern "C" int trx_launch_pack_trxd_wire(const trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
					 const float *d_soft, int soft_stride, const trxhip_trxd_meta *d_meta, uint8_t *d_pkt,
					 int pkt_stride, uint16_t *d_pkt_len, size_t n_bursts, float rssi_offset, hipStream_t stream,
					 trxhip_burst_result *d_results_copy)
{
	if (n_bursts == 0)
		return 0;
	if ((pkt_stride & 15) == 0 && n_bursts * (size_t)(pkt_stride >> 4) < 0xffffff00ull && ((uintptr_t)d_pkt & 15) == 0) {
		size_t blocks16 = (n_bursts * (size_t)(pkt_stride >> 4) + 255) / 256;
		if (blocks16 > 256 * 16) blocks16 = 256 * 16;
		hipLaunchKernelGGL(pack_trxd_wire16_kernel, dim3((unsigned)blocks16), dim3(256), 0, stream, d_results, d_params, d_soft,
				   soft_stride, d_meta, d_pkt, pkt_stride, d_pkt_len, (unsigned)n_bursts, rssi_offset, d_results_copy);
		return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
	}
	size_t blocks = (n_bursts * (size_t)(pkt_stride >> 2) + 255) / 256;
	if (blocks > 256 * 8) blocks = 256 * 8;
	hipLaunchKernelGGL(pack_trxd_wire_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_results, d_params, d_soft,
			   soft_stride, d_meta, d_pkt, pkt_stride, d_pkt_len, n_bursts, rssi_offset, d_results_copy);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// energyDetect() as a stand-alone call (sigProcLib.cpp:1573-1585): one wave per burst, mean |x|^2 over
// `window` samples taken at stride 4 from sample 0 (window clamped to the burst length as the reference does)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
energy_detect_kernel(const c32 *__restrict__ x, size_t n_bursts, int burst_len, unsigned window, float *__restrict__ out)
{
	const int lane = threadIdx.x & 63;
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	if (window > (unsigned)burst_len) window = burst_len;
	for (size_t b = wave; b < n_bursts; b += nwaves) {
		float e = 0.0f;
		for (unsigned i = lane; i < window; i += 64) {
			const c32 v = x[b * (size_t)burst_len + 4 * (size_t)i];
			e += v.y * v.y + v.x * v.x;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			e += __shfl_xor(e, o, 64);
		if (lane == 0)
			out[b] = window ? e / (float)window : 0.0f;
	}
}

extern "C" int trx_launch_energy_detect(const float *d_x, size_t n_bursts, int burst_len, unsigned window, float *d_out,
					hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	size_t blocks = (n_bursts + 3) / 4;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(energy_detect_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
			   reinterpret_cast<const c32 *>(d_x), n_bursts, burst_len, window, d_out);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// ------------------------------------------------------------------------------------------------
// Diversity-path selection of pullRadioVector() (Transceiver.cpp:723-741): a burst arrives on n_paths receive paths;
//   for each path i: pow = energyDetect(path i, 20 * sps); the FIRST path with the highest energy is demodulated
//   ("if (pow > max)" from max = -1, path order); avg += pow;  avg = sqrt(avg / chans) feeds rssi and the noise average.
// The selection is a decision, so each path's energy is the reference's serial sum (:1573-1585: energy += norm2 in
// sample order, one division) evaluated by one lane per path; then the whole wave copies the chosen path into the compact
// [n][burst_len] array the detector reads.  One wave per burst.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
diversity_select_kernel(const uint32_t *__restrict__ iq_paths, size_t n_bursts, int n_paths, int burst_len, unsigned window,
			uint32_t *__restrict__ iq_sel, float *__restrict__ avg_energy, uint8_t *__restrict__ path_out)
{
	const int lane = threadIdx.x & 63;
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	if (window > (unsigned)burst_len) window = burst_len;
	for (size_t b = wave; b < n_bursts; b += nwaves) {
		const uint32_t *src = iq_paths + b * (size_t)n_paths * burst_len;
		float pow = 0.0f;
		if (lane < n_paths) {
			const uint32_t *x = src + (size_t)lane * burst_len;
			float e = 0.0f;
			for (unsigned i = 0; i < window; i++) {
				const uint32_t w = x[4 * (size_t)i];
				const float re = (float)(int16_t)(w & 0xffffu), im = (float)(int16_t)(w >> 16);
				e += im * im + re * re;                                 /* Complex.h:113 norm2(): i*i + r*r */
			}
			pow = window ? e / (float)window : 0.0f;
		}
		float mx = -1.0f, avg = 0.0f;
		int sel = -1;
		for (int i = 0; i < n_paths; i++) {                                 /* wave-uniform: readlane per path */
			const float pi = __shfl(pow, i, 64);
			if (pi > mx) { mx = pi; sel = i; }
			avg += pi;
		}
		if (sel < 0) sel = 0;                                                   /* (cannot happen for finite input: pow >= 0 > -1) */
		const uint32_t *from = src + (size_t)sel * burst_len;
		uint32_t *to = iq_sel + b * (size_t)burst_len;
		for (int i = lane; i < burst_len; i += 64)
			to[i] = from[i];
		if (lane == 0) {
			avg_energy[b] = avg / (float)n_paths;                           /* avg^2 of Transceiver.cpp:741 */
			if (path_out) path_out[b] = (uint8_t)sel;
		}
	}
}

// result records of a detect/demod launch over the selected paths: energy and rssi from the path average
// (Transceiver.cpp:741,751: avg = sqrt(sum pow / chans); rssi = 20 log10(rxFullScale / avg)); OFF slots keep their zeros
__global__ void __launch_bounds__(256)
diversity_power_kernel(trxhip_burst_result *__restrict__ res, const trxhip_burst_params *__restrict__ params,
		       const float *__restrict__ avg_energy, size_t n_bursts, float full_scale)
{
	for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < n_bursts; b += (size_t)gridDim.x * blockDim.x) {
		if (params[b].type == TRXHIP_OFF)
			continue;
		const float e = avg_energy[b];
		res[b].energy = e;
		res[b].rssi = 6.02059991f * __log2f(full_scale) - 3.01029996f * __log2f(e);
	}
}

extern "C" int trx_launch_diversity_select(const int16_t *d_iq_paths, size_t n_bursts, int n_paths, int burst_len, int sps,
					   int16_t *d_iq_sel, float *d_avg_energy, uint8_t *d_path, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	size_t blocks = (n_bursts + 3) / 4;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(diversity_select_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
			   reinterpret_cast<const uint32_t *>(d_iq_paths), n_bursts, n_paths, burst_len, (unsigned)(20 * sps),
			   reinterpret_cast<uint32_t *>(d_iq_sel), d_avg_energy, d_path);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

extern "C" int trx_launch_diversity_power(trxhip_burst_result *d_res, const trxhip_burst_params *d_params, const float *d_avg_energy,
					  size_t n_bursts, float full_scale, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	size_t blocks = (n_bursts + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(diversity_power_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_res, d_params, d_avg_energy,
			   n_bursts, full_scale);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// vectorSlicer() (sigProcLib.cpp:546-556): dest = clamp(0.5 * (src + 1), 0, 1)
__global__ void __launch_bounds__(256)
vector_slicer_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t len)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = __builtin_amdgcn_fmed3f(0.5f * (src[i] + 1.0f), 0.0f, 1.0f);
}

extern "C" int trx_launch_vector_slicer(float *d_dst, const float *d_src, size_t len, hipStream_t stream)
{
	if (len == 0)
		return 0;
	size_t blocks = (len + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(vector_slicer_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_dst, d_src, len);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// delayVector() (sigProcLib.cpp:1046-1098) for a batch of equally long vectors, one delay per vector:
//   whole = floor(d), frac = d - whole; |frac| > 0.01: fshift[m] = sum_k X(m - 9 + k) * h_f[k], f = floorf(frac * 64),
//   20 real taps (convolve NO_DELAY, zero-padded :318-323), else fshift = x; out[i] = fshift[i - whole], 0 outside.
// One workgroup per vector: the vector is staged once in LDS between two zero pads (round 3: every output used to fetch its
// 20 samples from global memory -- 20 vector-memory instructions per output, 0.47 ms per 65536 x 625 samples), outputs are
// then computed from consecutive LDS entries (conflict-free) and stored coalesced; taps are wave-uniform (scalar loads), sums
// in tap order.  Vectors too long for the LDS take the direct form.
#define DV_PAD 10                                                       // X(m - 9 + k), k = 0..19: 9 before, 10 behind
__global__ void __launch_bounds__(256)
delay_vector_kernel(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ delays,
		    const trx_tables *__restrict__ tab, int len)
{
	extern __shared__ __attribute__((aligned(16))) char dv_smem[];
	c32 *xs = reinterpret_cast<c32 *>(dv_smem);                          // xs[DV_PAD + j] = x[j], zeros either side
	const size_t v = blockIdx.x;
	const c32 *x = in + v * (size_t)len;
	c32 *y = out + v * (size_t)len;
	const float delay = delays[v];
	const float fl = floorf(delay);
	const int whole = (int)fl;
	const float frac = delay - (float)whole;
	const bool use_filt = (double)fabsf(frac) > 1e-2;                  // :1056
	const int fidx = use_filt ? (int)floorf(frac * (float)TRX_DELAY_FILTS) : 0;
	const float *h = tab->delay_filt[fidx];
	for (int j = threadIdx.x; j < len + 2 * DV_PAD; j += blockDim.x) {
		const int jj = j - DV_PAD;
		xs[j] = (jj >= 0 && jj < len) ? x[jj] : make_float2(0.0f, 0.0f);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < len; i += blockDim.x) {
		const int m = i - whole;
		c32 r = make_float2(0.0f, 0.0f);
		if (m >= 0 && m < len) {
			if (use_filt) {
				const c32 *xp = xs + (m - 9 + DV_PAD);
				float yr = 0.0f, yi = 0.0f;
#pragma unroll
				for (int k = 0; k < TRX_DELAY_HLEN; k++) {
					const c32 xv = xp[k];
					yr += xv.x * h[k];
					yi += xv.y * h[k];
				}
				r = make_float2(yr, yi);
			} else {
				r = xs[m + DV_PAD];
			}
		}
		y[i] = r;
	}
}

// the direct form (any length): one thread per output, samples from global memory
__global__ void __launch_bounds__(256)
delay_vector_long_kernel(const c32 *__restrict__ in, c32 *__restrict__ out, const float *__restrict__ delays,
			 const trx_tables *__restrict__ tab, int len)
{
	const size_t v = blockIdx.y;
	const c32 *x = in + v * (size_t)len;
	c32 *y = out + v * (size_t)len;
	const float delay = delays[v];
	const float fl = floorf(delay);
	const int whole = (int)fl;
	const float frac = delay - (float)whole;
	const bool use_filt = (double)fabsf(frac) > 1e-2;                  // :1056
	const int fidx = use_filt ? (int)floorf(frac * (float)TRX_DELAY_FILTS) : 0;
	const float *h = tab->delay_filt[fidx];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
		const int m = i - whole;
		c32 r = make_float2(0.0f, 0.0f);
		if (m >= 0 && m < len) {
			if (use_filt) {
				float yr = 0.0f, yi = 0.0f;
#pragma unroll
				for (int k = 0; k < TRX_DELAY_HLEN; k++) {
					const int j = m - 9 + k;
					const c32 xv = (j >= 0 && j < len) ? x[j] : make_float2(0.0f, 0.0f);
					yr += xv.x * h[k];
					yi += xv.y * h[k];
				}
				r = make_float2(yr, yi);
			} else {
				r = x[m];
			}
		}
		y[i] = r;
	}
}

extern "C" int trx_launch_delay_vector(const float *d_in, float *d_out, const float *d_delays, const trx_tables *d_tab,
				       size_t n_vec, int len, hipStream_t stream)
{
	if (n_vec == 0 || len == 0)
		return 0;
	if (len <= 4096 && n_vec <= 0x7fffffffu) {                         // one workgroup per vector, the vector in LDS (<= 32 KB)
		const size_t lds = (size_t)(len + 2 * DV_PAD) * sizeof(c32);
		hipLaunchKernelGGL(delay_vector_kernel, dim3((unsigned)n_vec), dim3(256), lds, stream, reinterpret_cast<const c32 *>(d_in),
				   reinterpret_cast<c32 *>(d_out), d_delays, d_tab, len);
		return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
	}
	unsigned bx = (unsigned)((len + 255) / 256);
	if (bx > 64) bx = 64;
	for (size_t v0 = 0; v0 < n_vec; v0 += 65535) {                     // gridDim.y limit
		const size_t nv = (n_vec - v0 < 65535) ? n_vec - v0 : 65535;
		hipLaunchKernelGGL(delay_vector_long_kernel, dim3(bx, (unsigned)nv), dim3(256), 0, stream,
				   reinterpret_cast<const c32 *>(d_in) + v0 * (size_t)len,
				   reinterpret_cast<c32 *>(d_out) + v0 * (size_t)len, d_delays + v0, d_tab, len);
	}
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

// scaleVector() (sigProcLib.cpp:1188-1213): x[i] = x[i] * scale, Complex.h:74 operand order; in place
__global__ void __launch_bounds__(256)
scale_vector_kernel(c32 *__restrict__ x, size_t len, c32 scale)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
	{
		const c32 a = x[i];
		x[i] = make_float2(a.x * scale.x - a.y * scale.y, a.x * scale.y + a.y * scale.x);
	}
}

extern "C" int trx_launch_scale_vector(float *d_x, size_t len, float sr, float si, hipStream_t stream)
{
	if (len == 0)
		return 0;
	size_t blocks = (len + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(scale_vector_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<c32 *>(d_x), len,
			   make_float2(sr, si));
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}


// ---- bursts by reference (trxhip_hostpipe_submit_by_ref): fetch n bursts through n pointers into one dense block ----
// src[b] is the device-side address of burst b in host memory that was pinned and mapped (hipHostRegister): the loads cross
// the link.  One wave per burst, `dwords` 4-byte words each (625 at 4 SPS: ten rounds of 256 contiguous bytes); five loads of a
// lane are issued before the first store so that a wave keeps 1.25 KB in flight -- 256 CUs x 8 waves hold 2.6 MB on the link,
// an order of magnitude above its bandwidth-delay product.  Plain loads: with the non-temporal hint the same kernel fetched
// 37.7 GB/s instead of 44.7, 16-byte loads from the aligned body of every burst 41.4 / 42.8 (with / without the hint;
// profiles/r05_gather.txt; the copy engine
// moves the staged form of the same batch at 55 GB/s -- a cache line per request against the engine's long bursts).
// Nothing of this data is read twice: HBM sees one write.
__global__ void __launch_bounds__(256)
gather_bursts_kernel(const unsigned long long *__restrict__ src, uint32_t *__restrict__ dst, size_t n, unsigned dwords)
{
	const unsigned lane = threadIdx.x & 63u;
	const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (b >= n)
		return;
	const unsigned long long a = src[b];                                   // wave-uniform: a scalar base, 32-bit lane offsets
	if (a == 0ull)                                                         // this burst came with a run the copy engine moved
		return;
	const unsigned a_lo = __builtin_amdgcn_readfirstlane((unsigned)a), a_hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	const uint32_t *__restrict__ s = reinterpret_cast<const uint32_t *>(((unsigned long long)a_hi << 32) | a_lo);
	uint32_t *__restrict__ d = dst + b * dwords;
	unsigned i = lane;
	for (; i + 4u * 64u < dwords; i += 5u * 64u) {
		uint32_t v[5];
#pragma unroll
		for (int k = 0; k < 5; k++)
			v[k] = s[i + 64u * k];
#pragma unroll
		for (int k = 0; k < 5; k++)
			d[i + 64u * k] = v[k];
	}
	for (; i < dwords; i += 64u)
		d[i] = s[i];
}

extern "C" int trx_launch_gather_bursts(const unsigned long long *d_src, void *d_dst, size_t n, unsigned dwords, hipStream_t stream)
{
	if (n == 0)
		return 0;
	hipLaunchKernelGGL(gather_bursts_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, d_src, reinterpret_cast<uint32_t *>(d_dst), n, dwords);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
