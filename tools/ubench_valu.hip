// ubench_valu.hip -- issue cost of the instruction kinds the burst kernels are made of, on one CU of gfx950,
// for 1 / 2 / 4 waves per SIMD (the production kernel runs 4).  Stand-alone: hipcc --offload-arch=gfx950 -O3.
//   cycles per instruction per SIMD = elapsed shader cycles / (instructions per wave x waves per SIMD)
// Output: one line per (kind, waves per SIMD).  Used to price instruction-mix changes before building them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

enum Kind { K_FMA, K_PKFMA, K_PKADD, K_PKMUL, K_ADD, K_MAX3, K_CVT_SDWA, K_LOG, K_RCP, K_DPP_ADD_DEP, K_DPP_MOV_INDEP,
	    K_LDS_B64, K_LDS_B64_BCAST, K_LDS_B128_BCAST, K_LDS_B32, K_SALU, K_MIX_PK_LDS, K_MIX_PK_SALU, K_MIX_FMA_PKFMA,
	    K_READLANE, K_WRITE_B64, K_FMA_DEP, K_PKFMA_DEP, K_MIX3, K_NKINDS };
static const char *kind_name[] = { "v_fma_f32 x8 indep", "v_pk_fma_f32 x8 indep", "v_pk_add_f32 x8 indep", "v_pk_mul_f32 x8 indep",
	"v_add_f32 x8 indep", "v_max3_f32 |a|,|b| x8", "v_cvt_f32_i32_sdwa x8", "v_log_f32 x8", "v_rcp_f32 x8",
	"s_nop1 + v_add_f32_dpp dependent", "v_mov_b32_dpp indep x8", "ds_read_b64 stride-8B x8", "ds_read_b64 broadcast x8",
	"ds_read_b128 broadcast x8", "ds_read_b32 stride-4B x8", "s_add_u32 x8", "mix: pk_fma + ds_read_b64 (1:1)",
	"mix: pk_fma + s_add (1:1)", "mix: v_fma + v_pk_fma (1:1)", "v_readlane_b32 x8", "ds_write_b64 x8", "v_fma_f32 dependent chain",
	"v_pk_fma_f32 dependent chain", "mix: pk_fma + ds_read_b64 + s_add (1:1:1)" };

template <int KIND>
__global__ void __launch_bounds__(1024) ub(unsigned long long *out, int iters, float seed)
{
	__shared__ __attribute__((aligned(16))) float lds[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
	v2f p0 = { seed, seed }, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
	v2f x = { seed * 0.5f, seed * 0.25f }, h = { 1.0f + seed, 1.0f - seed };
	float4 q0, q1, q2, q3, q4, q5, q6, q7;
	unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
	const unsigned a64 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds + lane * 8;
	const unsigned a32 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds + lane * 4;
	const unsigned ab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
	int ri = (int)seed;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
		if (KIND == K_FMA)
			asm volatile(REP16("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
					   "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x.x), "v"(h.x));
		if (KIND == K_FMA_DEP)
			asm volatile(REP64("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n") : "+v"(a0) : "v"(x.x), "v"(h.x));
		if (KIND == K_PKFMA_DEP)
			asm volatile(REP64("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n") : "+v"(p0) : "v"(x), "v"(h));
		if (KIND == K_ADD)
			asm volatile(REP16("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
					   "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x.x));
		if (KIND == K_MAX3)
			asm volatile(REP16("v_max3_f32 %0, %0, |%8|, |%9|\n v_max3_f32 %1, %1, |%8|, |%9|\n v_max3_f32 %2, %2, |%8|, |%9|\n v_max3_f32 %3, %3, |%8|, |%9|\n"
					   "v_max3_f32 %4, %4, |%8|, |%9|\n v_max3_f32 %5, %5, |%8|, |%9|\n v_max3_f32 %6, %6, |%8|, |%9|\n v_max3_f32 %7, %7, |%8|, |%9|\n")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x.x), "v"(h.x));
		if (KIND == K_CVT_SDWA)
			asm volatile(REP16("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
					   "v_cvt_f32_i32_sdwa %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
					   "v_cvt_f32_i32_sdwa %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
					   "v_cvt_f32_i32_sdwa %6, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %7, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
				     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(ri));
		if (KIND == K_LOG)
			asm volatile(REP16("v_log_f32 %0, %8\n v_log_f32 %1, %8\n v_log_f32 %2, %8\n v_log_f32 %3, %8\n"
					   "v_log_f32 %4, %8\n v_log_f32 %5, %8\n v_log_f32 %6, %8\n v_log_f32 %7, %8\n")
				     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(x.x));
		if (KIND == K_RCP)
			asm volatile(REP16("v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n"
					   "v_rcp_f32 %4, %8\n v_rcp_f32 %5, %8\n v_rcp_f32 %6, %8\n v_rcp_f32 %7, %8\n")
				     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(x.x));
		if (KIND == K_PKFMA)
			asm volatile(REP16("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
					   "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(x), "v"(h));
		if (KIND == K_PKADD)
			asm volatile(REP16("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n"
					   "v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(x));
		if (KIND == K_PKMUL)
			asm volatile(REP16("v_pk_mul_f32 %0, %8, %9\n v_pk_mul_f32 %1, %8, %9\n v_pk_mul_f32 %2, %8, %9\n v_pk_mul_f32 %3, %8, %9\n"
					   "v_pk_mul_f32 %4, %8, %9\n v_pk_mul_f32 %5, %8, %9\n v_pk_mul_f32 %6, %8, %9\n v_pk_mul_f32 %7, %8, %9\n")
				     : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7) : "v"(x), "v"(h));
		if (KIND == K_MIX_FMA_PKFMA)
			asm volatile(REP16("v_pk_fma_f32 %0, %8, %9, %0\n v_fma_f32 %4, %10, %11, %4\n v_pk_fma_f32 %1, %8, %9, %1\n v_fma_f32 %5, %10, %11, %5\n"
					   "v_pk_fma_f32 %2, %8, %9, %2\n v_fma_f32 %6, %10, %11, %6\n v_pk_fma_f32 %3, %8, %9, %3\n v_fma_f32 %7, %10, %11, %7\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(h), "v"(x.x), "v"(h.x));
		if (KIND == K_DPP_ADD_DEP)
			asm volatile(REP64("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
				     : "+v"(a0) : "v"(x.x));
		if (KIND == K_DPP_MOV_INDEP)
			asm volatile(REP16("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					   "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					   "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					   "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x.x));
		if (KIND == K_READLANE)
			asm volatile(REP16("v_readlane_b32 %0, %8, 5\n v_readlane_b32 %1, %8, 6\n v_readlane_b32 %2, %8, 7\n v_readlane_b32 %3, %8, 8\n"
					   "v_readlane_b32 %4, %8, 9\n v_readlane_b32 %5, %8, 10\n v_readlane_b32 %6, %8, 11\n v_readlane_b32 %7, %8, 12\n")
				     : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(x.x));
		if (KIND == K_LDS_B64)
			asm volatile(REP16("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
					   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(4)\n")
				     : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7) : "v"(a64) : "memory");
		if (KIND == K_LDS_B32)
			asm volatile(REP16("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:512\n ds_read_b32 %2, %8 offset:1024\n ds_read_b32 %3, %8 offset:1536\n"
					   "ds_read_b32 %4, %8 offset:2048\n ds_read_b32 %5, %8 offset:2560\n ds_read_b32 %6, %8 offset:3072\n ds_read_b32 %7, %8 offset:3584\n s_waitcnt lgkmcnt(4)\n")
				     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(a32) : "memory");
		if (KIND == K_LDS_B64_BCAST)
			asm volatile(REP16("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
					   "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(4)\n")
				     : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7) : "v"(ab) : "memory");
		if (KIND == K_LDS_B128_BCAST)
			asm volatile(REP16("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
					   "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(4)\n")
				     : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(ab) : "memory");
		if (KIND == K_WRITE_B64)
			asm volatile(REP16("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
					   "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n s_waitcnt lgkmcnt(4)\n")
				     : : "v"(a64), "v"(x) : "memory");
		if (KIND == K_SALU)
			asm volatile(REP16("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
					   "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n")
				     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
		if (KIND == K_MIX_PK_LDS)
			asm volatile(REP16("v_pk_fma_f32 %0, %8, %9, %0\n ds_read_b64 %4, %10\n v_pk_fma_f32 %1, %8, %9, %1\n ds_read_b64 %5, %10 offset:512\n"
					   "v_pk_fma_f32 %2, %8, %9, %2\n ds_read_b64 %6, %10 offset:1024\n v_pk_fma_f32 %3, %8, %9, %3\n ds_read_b64 %7, %10 offset:1536\n s_waitcnt lgkmcnt(2)\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7) : "v"(x), "v"(h), "v"(a64) : "memory");
		if (KIND == K_MIX_PK_SALU)
			asm volatile(REP16("v_pk_fma_f32 %0, %8, %9, %0\n s_add_u32 %4, %4, 3\n v_pk_fma_f32 %1, %8, %9, %1\n s_add_u32 %5, %5, 3\n"
					   "v_pk_fma_f32 %2, %8, %9, %2\n s_add_u32 %6, %6, 3\n v_pk_fma_f32 %3, %8, %9, %3\n s_add_u32 %7, %7, 3\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(x), "v"(h) : "scc");
		if (KIND == K_MIX3)
			asm volatile(REP16("v_pk_fma_f32 %0, %10, %11, %0\n ds_read_b64 %4, %12\n s_add_u32 %8, %8, 3\n v_pk_fma_f32 %1, %10, %11, %1\n ds_read_b64 %5, %12 offset:512\n s_add_u32 %9, %9, 3\n"
					   "v_pk_fma_f32 %2, %10, %11, %2\n ds_read_b64 %6, %12 offset:1024\n s_add_u32 %8, %8, 3\n v_pk_fma_f32 %3, %10, %11, %3\n ds_read_b64 %7, %12 offset:1536\n s_add_u32 %9, %9, 3\n s_waitcnt lgkmcnt(2)\n")
				     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=v"(p4), "=v"(p5), "=v"(p6), "=v"(p7), "+s"(s0), "+s"(s1) : "v"(x), "v"(h), "v"(a64) : "memory", "scc");
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	const unsigned long long t1 = __builtin_readcyclecounter();
	float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y +
		     q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
	if (sink == 12345.678f) out[1023] = 1;
	if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(unsigned long long *d_out, int waves_per_simd)
{
	const int iters = 64;
	const int threads = 64 * 4 * waves_per_simd;
	hipLaunchKernelGGL(ub<KIND>, dim3(1), dim3(threads), 0, 0, d_out, iters, 0.001f);     // warm-up
	hipLaunchKernelGGL(ub<KIND>, dim3(1), dim3(threads), 0, 0, d_out, iters, 0.001f);
	hipDeviceSynchronize();
	std::vector<unsigned long long> h(16);
	hipMemcpy(h.data(), d_out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	unsigned long long mx = 0;
	for (int w = 0; w < 4 * waves_per_simd; w++) mx = h[w] > mx ? h[w] : mx;
	// instructions per wave per iteration: 128 of the measured kind(s) (mixes: 128 + 128 (+128); dependent DPP: 128 pairs)
	const double per = (double)mx / (double)iters / 128.0 / (double)waves_per_simd;
	printf("%-44s waves/SIMD %d : %7.2f cycles per (group of) instruction per SIMD   (wave elapsed %llu)\n", kind_name[KIND], waves_per_simd, per, mx);
}

template <int KIND>
static void run_all(unsigned long long *d_out)
{
	run<KIND>(d_out, 1);
	run<KIND>(d_out, 2);
	run<KIND>(d_out, 4);
}

int main()
{
	unsigned long long *d_out;
	if (hipMalloc(&d_out, 1024 * sizeof(unsigned long long)) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
	run_all<K_FMA>(d_out); run_all<K_FMA_DEP>(d_out); run_all<K_PKFMA>(d_out); run_all<K_PKFMA_DEP>(d_out); run_all<K_PKADD>(d_out); run_all<K_PKMUL>(d_out); run_all<K_ADD>(d_out);
	run_all<K_MIX_FMA_PKFMA>(d_out);
	run_all<K_MAX3>(d_out); run_all<K_CVT_SDWA>(d_out); run_all<K_LOG>(d_out); run_all<K_RCP>(d_out);
	run_all<K_DPP_ADD_DEP>(d_out); run_all<K_DPP_MOV_INDEP>(d_out); run_all<K_READLANE>(d_out);
	run_all<K_LDS_B64>(d_out); run_all<K_LDS_B64_BCAST>(d_out); run_all<K_LDS_B128_BCAST>(d_out); run_all<K_LDS_B32>(d_out); run_all<K_WRITE_B64>(d_out);
	run_all<K_SALU>(d_out); run_all<K_MIX_PK_LDS>(d_out); run_all<K_MIX_PK_SALU>(d_out); run_all<K_MIX3>(d_out);
	hipFree(d_out);
	return 0;
}
