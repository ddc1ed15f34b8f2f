// trx_k4_common.h -- pieces shared by the 4-SPS kernels (trx_kernel4.hip: every burst type; trx_kernel_nb.hip: normal bursts only):
// the polyphase LDS layout of a burst, the /4 decimator and the fused demodulator's main filter.
#pragma once
#include "trx_device.h"

#define PH_A   180                 // entries per phase array (= 4 mod 16: conflict-free loader writes)
#define PH_M0  12                  // position of m = 0 inside a phase array (48 samples of zero pad in front)
#define K4_XS  (4 * PH_A)

typedef float v2f __attribute__((ext_vector_type(2)));

// four per-lane base pointers for a run of consecutive samples s0, s0+1, ... : sample s0+t lives at
// pb[t & 3][t >> 2].  ph0 = s0 & 3, m0 = s0 >> 2 (arithmetic).
struct PhBase { const c32 *p[4]; };
__device__ __forceinline__ PhBase ph_bases(const c32 *P, int ph0, int m0)
{
	// p[k] = P + ((ph0 + k) & 3) * PH_A + PH_M0 + m0 + ((ph0 + k) >> 2), incrementally: one multiply-add for p[0],
	// then + k * PH_A, and one row back / one sample on where the phase wraps
	PhBase b;
	const c32 *p0 = P + (ph0 * PH_A + PH_M0 + m0);
	b.p[0] = p0;
#pragma unroll
	for (int k = 1; k < 4; k++)
		b.p[k] = p0 + (k * PH_A + ((ph0 + k) >> 2) * (1 - 4 * PH_A));
	return b;
}

// The fused demodulator's main filter: three ADJACENT outputs per lane over the composite taps u = K4_U0 .. K4_U0 + K4_NT - 1
// (6 .. 29, 24 of 35: trx_tables.h).  pb addresses the lane's first sample (tap K4_U0 of its first output), c4 its tap row from K4_U0 on -- a per-lane
// LDS address, wave-uniform for the ordinary lanes.  Taps outer, a ring of 16 samples loaded D ahead of use; one
// sched_barrier per tap keeps the order and the register footprint.
#define K4_U0 TRX_FUSED_U0
#define K4_NT TRX_FUSED_NT
#define K4_NTP TRX_FUSED_NTP
__device__ __forceinline__ void fir24x3(const PhBase &pb, const float4 *c4, v2f (&acc)[3])
{
	constexpr int D = 4, NV = K4_NT + 8;                            // samples v = 0 .. 31; the ring runs D = 4 samples ahead of the FMAs
	c32 xw[16];
	float4 cq[2];
	cq[0] = c4[0];
#pragma unroll
	for (int v = 0; v < 8 + D; v++)
		xw[v] = lds_c32(pb.p[v & 3] + (v >> 2));
#pragma unroll
	for (int u = 0; u < K4_NT; u++) {
		if ((u & 3) == 0 && u + 4 < K4_NT)
			cq[((u >> 2) + 1) & 1] = c4[(u >> 2) + 1];
		// the ring is refilled four samples at a time, in front of every group of four taps: the compiler then needs ONE
		// s_waitcnt per group (for the previous group's reads) where a read per tap needed one per tap -- a wait is an issue
		// slot like any other (round 5: 24 -> 6 in this filter, - 0.9 % wave cycles)
		if ((u & 3) == 0) {
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (u + 8 + D + k < NV)
					xw[(u + 8 + D + k) & 15] = lds_c32(pb.p[(u + 8 + D + k) & 3] + ((u + 8 + D + k) >> 2));
		}
		const float4 ca = cq[(u >> 2) & 1];
		const v2f hp = (u & 2) ? (v2f){ ca.z, ca.w } : (v2f){ ca.x, ca.y };
#pragma unroll
		for (int j = 0; j < 3; j++) {
			const v2f xv = { xw[(u + 4 * j) & 15].x, xw[(u + 4 * j) & 15].y };
			acc[j] = (u & 1) ? pk_fma_tap<1>(xv, hp, acc[j]) : pk_fma_tap<0>(xv, hp, acc[j]);
		}
		__builtin_amdgcn_sched_barrier(0);
	}
}

// One output of the /4 decimator (downsampleBurst, :1587-1601) on the polyphase layout: y = sum_k x[4i-15+k] * g[k], product
// then sum, k ascending (the reference's order).  All 16 samples are fetched before the first multiply -- the compiler's own
// schedule interleaves reads and waits with 2-5 reads in flight and exposes the LDS latency six times -- and the taps come as
// g[0..7] only (two 16-byte broadcast reads): the filter is bitwise symmetric, g[k] == g[15-k] (checked when the context
// is created; TRX_IFLAG_NO_SYM otherwise keeps callers on the generic path).
//   FROM_P0: start the sum at the first product instead of adding it to +0 (one instruction less; differs from the reference
//   only in the sign of a zero result when every product is -0: the fused kernels take it, the bit-exact ones do not)
template <bool FROM_P0 = false>
__device__ __forceinline__ c32 decimate16_sym(const c32 *pd, const float *gdec)
{
	const float4 *g4 = reinterpret_cast<const float4 *>(gdec);
	c32 xs[16];
#pragma unroll
	for (int k = 0; k < 16; k++)
		xs[k] = lds_c32(pd + ((k + 1) & 3) * PH_A + ((k + 1) >> 2));
	const float4 gA = g4[0], gB = g4[1];
	__builtin_amdgcn_sched_barrier(0);
	// product k + 1 is issued between sum k - 1 and sum k: a v_pk_add_f32 straight behind the v_pk_add_f32 it depends on costs
	// a wait state each time (the compiler's order -- sixteen products, then sixteen sums -- paid fifteen s_nop)
	auto prod = [&](int k) {
		const int kk = k < 8 ? k : 15 - k;
		const float4 gq = (kk >> 2) ? gB : gA;
		const v2f gp = (kk & 2) ? (v2f){ gq.z, gq.w } : (v2f){ gq.x, gq.y };
		const v2f xv = { xs[k].x, xs[k].y };
		v2f r;
		if (kk & 1)
			asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(xv), "v"(gp));
		else
			asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(xv), "v"(gp));
		return r;
	};
	v2f p_cur = prod(0), p_next = prod(1);
	v2f ya;
	if (FROM_P0) {
		ya = p_cur;
	} else {
		ya = (v2f){ 0.0f, 0.0f };
		asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ya) : "v"(p_cur));
	}
#pragma unroll
	for (int k = 1; k < 16; k++) {
		p_cur = p_next;
		if (k + 1 < 16)
			p_next = prod(k + 1);
		asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ya) : "v"(p_cur));
	}
	return make_float2(ya.x, ya.y);
}
