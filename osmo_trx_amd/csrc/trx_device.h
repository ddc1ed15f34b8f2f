// trx_device.h -- device-side building blocks shared by the burst kernels (gfx950, wave64):
// wave helpers (DPP reductions, ballots), the sinc-LUT interpolation, the speculative TOA bisection and
// detectBurst().  See trx_kernels.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "trx_tables.h"
#include "../../include/trxhip.h"

#define WAVE 64
#define TRX_PAD 20                 // zero samples kept on both sides of a burst in LDS
#define TRX_DEC_LEN 208            // decimated burst: 156 samples + zero tail up to start+len (<= 199)
#define TRX_CORR_MAX 128           // head + tail <= 16 + TRXHIP_MAX_TOA
// narrow per-wave buffers of the 4-SPS production kernel (16 waves per CU must fit 160 KB of LDS): windows up to
// max_toa = 64 (the reference's defaults are 63 / 30) are stored whole, wider ones take detect_burst()'s windowed path
#define TRX_DEC_NARROW 160         // 156 samples + 4 zeros
#define TRX_CORR_NARROW 80
#define TRX_CZ_PAD 12              // zero samples either side of the correlation (interpolatePoint reach)
#define TRX_CZ_LEN (TRX_CZ_PAD + TRX_CORR_MAX + TRX_CZ_PAD)
#define TRX_SINCV_LDS (TRX_SINCV_LEN + 32)   // + zero tail: q = 4096 is addressed when the fraction is 0
#define TRX_CLIP_THRESH 30000.0f   // sigProcLib.cpp:49
#define TRX_WPB 12                 // waves (= bursts in flight) per workgroup; one workgroup per CU

// LDS-resident sequence table: [8 TSC x 16][3 RACH x 40][8 EDGE x 16] taps, then 19 headers of 8 floats
#define LSEQ_TSC(s)   ((s) * 16)
#define LSEQ_RACH(i)  (128 + (i) * 40)
#define LSEQ_EDGE(s)  (248 + (s) * 16)
#define LSEQ_DUMMY    376                 /* gDummySequence, 16 taps */
#define LSEQ_TAPS     392
#define LSEQ_NHDR     20
#define TRX_TABLES_LDS_FLOATS (TRX_SINCV_LDS + TRX_DELAY_FILTS * TRX_DELAY_HLEN + 2 * 160 + 16 + 2 * LSEQ_TAPS + 8 * LSEQ_NHDR)
#define TRX_TABLES_LDS_BYTES (TRX_TABLES_LDS_FLOATS * 4)

typedef float2 c32;

// The > 64 KB dynamic-LDS opt-in is a per-kernel, per-device attribute: arm it ONCE with the kernel's maximum (160 KB)
// instead of the size of the current call -- concurrent callers with different sizes (the C ABI allows calls from
// several host threads) would otherwise race between one thread's hipFuncSetAttribute and another's launch.
#define TRX_ARM_DYNAMIC_LDS(kernel_ptr)                                                                          \
	do {                                                                                                     \
		static std::atomic<unsigned long long> armed_{0ull};                                             \
		int dev_ = 0;                                                                                    \
		if (hipGetDevice(&dev_) != hipSuccess) return TRXHIP_EIO;                                        \
		const unsigned long long bit_ = 1ull << (dev_ & 63);                                             \
		if (!(armed_.load(std::memory_order_acquire) & bit_)) {                                          \
			if (hipFuncSetAttribute((const void *)(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, \
						160 * 1024) != hipSuccess)                                        \
				return TRXHIP_EIO;                                                               \
			armed_.fetch_or(bit_, std::memory_order_release);                                        \
		}                                                                                                \
	} while (0)

// Diagnostic build only (-DTRX_DIAG, libtrxhip_diag.so): the upper bits of `slice` carry a phase-ablation
// mask so that per-phase cost can be measured on the GPU.  The product library is built without it.
#ifdef TRX_DIAG
#define ABL(bit) ((slice >> (8 + (bit))) & 1)
// per-phase cycle accounting (s_memtime deltas summed over all bursts by lane 0 of every wave)
#define TRX_DIAG_WAVES 8192
static __device__ unsigned long long g_trx_diag[TRX_DIAG_WAVES * 24];   // per wave (no atomics); one copy per translation unit
// slots 20..23 are not sums: wall-clock start / end of the wave (s_memrealtime, 100 MHz), HW_ID and XCC_ID
#define DIAG_DECL unsigned long long diag_acc[24] = {0}; unsigned long long diag_prev = __builtin_readcyclecounter(); \
	const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime()
#define DIAG_ARG , unsigned long long *diag_acc, unsigned long long &diag_prev
#define DIAG_PASS , diag_acc, diag_prev
#define DIAG_MARK(k)                                                                            \
	do {                                                                                    \
		const unsigned long long _t = __builtin_readcyclecounter();                     \
		diag_acc[k] += _t - diag_prev;                                                  \
		diag_prev = _t;                                                                 \
	} while (0)
#define DIAG_FLUSH()                                                                            \
	do {                                                                                    \
		const unsigned _w = (blockIdx.x * 16 + (threadIdx.x >> 6)) % TRX_DIAG_WAVES;    \
		if ((threadIdx.x & 63) == 0) {                                                  \
			for (int _k = 0; _k < 20; _k++) g_trx_diag[_w * 24 + _k] += diag_acc[_k]; \
			unsigned _hw, _xcc;                                                     \
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(_hw));        \
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(_xcc));      \
			g_trx_diag[_w * 24 + 20] = diag_t0;                                     \
			g_trx_diag[_w * 24 + 21] = __builtin_amdgcn_s_memrealtime();            \
			g_trx_diag[_w * 24 + 22] = _hw;                                         \
			g_trx_diag[_w * 24 + 23] = _xcc;                                        \
		}                                                                               \
	} while (0)
#else
// tools/pmc_phase_r03.sh: per-phase instruction counts of the PRODUCT kernel (the COMMON instantiation takes no run-time
// flags) come from measurement builds with a compile-time ablation mask; the product library is built without it
#ifdef TRX_ABL_MASK
#define ABL(bit) ((TRX_ABL_MASK >> (bit)) & 1)
#else
#define ABL(bit) 0
#endif
#define DIAG_DECL
#define DIAG_ARG
#define DIAG_PASS
#define DIAG_MARK(k)
#define DIAG_FLUSH()
#endif

// Measurement build only (-DTRX_WHATIF_PAIR, tools/build_variants.py): an UPPER BOUND on what packing two bursts into one
// wave for correlation / arg-max / the two gates / computeCI could buy.  Every other normal burst of a
// wave skips exactly those phases and reuses the previous burst's peak index -- as if the neighbour's pass had produced both
// results at no extra cost -- while decimation, both bisection rounds and the demodulator (phases whose lanes are full, or
// whose 34 + 34 samples do not fit 64 lanes) run as always.  Results are wrong by construction: timing and counters only.
#ifdef TRX_WHATIF_PAIR
struct WhatIf { int bidx; int skip; };
#define WI_ARG , WhatIf &wi
#define WI_PASS , wi
#define WI_LOCAL WhatIf wi = { 0, 0 }
#define WI_SKIP (wi.skip != 0)
#else
#define WI_ARG
#define WI_PASS
#define WI_LOCAL
#define WI_SKIP false
#endif

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
	// all 64 lanes of a wave execute LDS instructions in order: a compiler-level fence is enough to
	// make this wave's earlier LDS writes visible to its later reads from other lanes.
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// value held by lane `l` (l wave-uniform): v_readlane_b32
__device__ __forceinline__ float lane_val(float v, int l)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l)));
}

// v_writelane_b32: the wave-uniform value `sval` into lane LANE of `word`, the other lanes untouched
template <int LANE>
__device__ __forceinline__ int write_lane(int word, int sval)
{
	asm("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(__builtin_amdgcn_readfirstlane(sval)), "n"(LANE));
	return word;
}

// the (wave-uniform but vector-resident) value `vval` into lane LANE of `word`: one v_cndmask under a constant lane mask, where
// v_readfirstlane + v_writelane would be two vector instructions
template <int LANE>
__device__ __forceinline__ int put_lane(int word, float vval)
{
	const unsigned long long m = 1ull << LANE;
	asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(word) : "v"(vval), "s"(m));
	return word;
}

// DPP move: lanes without a valid source keep their own value
// Work claiming from a workgroup-wide LDS counter, split in two so that the atomic's latency is covered by whatever runs
// between issue and use: lane 0 alone executes one ds_add_rtn_u32 (exec is narrowed around it -- both calls sit in
// wave-uniform code where every lane is active), the other half waits for it and moves the ticket to a scalar register.
// The compiler's own form of "if (lane == 0) atomic" is the generic wave-aggregated sequence (mbcnt, bcnt, saveexec
// twice, ~9 VALU + ~12 SALU) with a full wait right behind the atomic.
__device__ __forceinline__ int claim_issue(const int *lds_counter)
{
	const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) int *)lds_counter;
	int ticket;
	asm volatile("s_mov_b64 exec, 1\n\t"
		     "ds_add_rtn_u32 %0, %1, %2\n\t"
		     "s_mov_b64 exec, -1"
		     : "=&v"(ticket) : "v"(addr), "v"(1) : "memory");
	return ticket;                                   // lane 0, once the LDS has answered
}
__device__ __forceinline__ int claim_take(int ticket)
{
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ticket) : : "memory");
	return __builtin_amdgcn_readfirstlane(ticket);
}

// the same when exactly >= N LDS instructions (and no scalar load) were issued behind claim_issue() on EVERY path that reaches
// the call: the LDS answers in order, so the ticket has arrived once at most N operations are outstanding -- the writes behind
// it need not have drained.  (N above the real count would return before the ticket has arrived: callers state a lower bound.)
template <int N>
__device__ __forceinline__ int claim_take_behind(int ticket)
{
	asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ticket) : "n"(N) : "memory");
	return __builtin_amdgcn_readfirstlane(ticket);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define DPP_QUAD_XOR1   0xB1   // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2   0x4E   // quad_perm:[2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR  0x140
#define DPP_BCAST15     0x142
#define DPP_BCAST31     0x143

// wave-wide max / sum in 6 DPP VALU ops; result taken from lane 63.  Written as asm: from the update_dpp builtin
// the compiler emits v_mov_b32 + v_mov_b32_dpp + (canonicalising v_max) + op per step (~4x the instructions).
// A DPP source written by the previous VALU op needs two wait states on gfx9 (s_nop 1), which the compiler's
// hazard recogniser does not insert inside an asm block.
#define TRX_DPP_STEP(op, ctrl) "s_nop 1\n\t" op " %0, %0, %0 " ctrl "\n\t"
__device__ __forceinline__ float wave_max(float v)
{
	asm volatile(TRX_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_max_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_max_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
		     TRX_DPP_STEP("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
		     : "+v"(v));
	return lane_val(v, 63);
}

__device__ __forceinline__ float wave_sum(float v)
{
	// rows 1,3 += row 0,2 totals; rows 2,3 += (rows 0+1) total: masked-out rows are not written
	asm volatile(TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
		     : "+v"(v));
	return lane_val(v, 63);
}

// wave max of `m` and wave sum of `s` in one go: the two DPP chains are interleaved, so each fills one of the two
// wait states the other needs and a single s_nop 0 per pair replaces two s_nop 1
__device__ __forceinline__ void wave_max_and_sum(float &m, float &s)
{
#define TRX_DPP_PAIR(ctrl) "s_nop 0\n\tv_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\t"
	asm volatile("s_nop 1\n\t"
		     TRX_DPP_PAIR("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_PAIR("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_PAIR("row_half_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_PAIR("row_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_PAIR("row_bcast:15 row_mask:0xa bank_mask:0xf")
		     TRX_DPP_PAIR("row_bcast:31 row_mask:0xc bank_mask:0xf")
		     : "+v"(m), "+v"(s));
#undef TRX_DPP_PAIR
	m = lane_val(m, 63);
	s = lane_val(s, 63);
}

// wave sum of the values held by the lanes = 0 mod 4 (what the other lanes hold is ignored): the quad's lane 0 is broadcast
// to its quad, then four DPP adds -- five instructions where zeroing the other lanes and the full tree take seven
__device__ __forceinline__ float wave_sum_quad0(float v)
{
	asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
		     TRX_DPP_STEP("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
		     : "+v"(v));
	return lane_val(v, 63);
}

// sum over each row of 16 lanes (every lane of the row gets the total)
__device__ __forceinline__ float row_sum(float v)
{
	asm volatile(TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
		     TRX_DPP_STEP("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
		     : "+v"(v));
	return v;
}

// One complex sample from LDS as its own ds_read_b64.  Left to itself the compiler pairs neighbouring 8-byte reads
// into ds_read2_b64, which occupies the LDS for 8 cycles where two ds_read_b64 take 4 (MI355X_MICROARCH.md, LDS
// table); the volatile qualifier only stops that merge, the reads still issue back to back.
typedef float trx_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c32 lds_c32(const c32 *p)
{
	typedef const volatile trx_v2f __attribute__((address_space(3))) *lds_ptr;   // explicit LDS pointer: a volatile
	const trx_v2f v = *(lds_ptr)(p);                                             // generic access would be a flat_load
	return make_float2(v.x, v.y);
}

// Complex.h:74 operator*(Complex): (a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x) as two packed multiplies and one packed add
// with the low half negated -- the same four roundings in 3 instructions (the compiler's own version takes 6)
__device__ __forceinline__ c32 cmul(c32 a, c32 b)
{
	const trx_v2f p = (trx_v2f){ a.x, a.x } * (trx_v2f){ b.x, b.y };
	const trx_v2f q = (trx_v2f){ a.y, a.y } * (trx_v2f){ b.y, b.x };
	trx_v2f t;
	asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(t) : "v"(p), "v"(q));
	return make_float2(t.x, t.y);
}

// acc += x * h for a complex x and a real tap h that is element HI of a 64-bit register pair (taps arrive from LDS
// in pairs / quads): one v_pk_fma_f32 with the tap selected by op_sel.  The compiler only knows how to broadcast
// the low element and spends a v_mov on every odd tap.
template <int HI>
__device__ __forceinline__ trx_v2f pk_fma_tap(trx_v2f x, trx_v2f hpair, trx_v2f acc)
{
	if (HI)
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(hpair));
	else
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(hpair));
	return acc;
}

// x * h for a complex x and a real tap h that is element HI of a 64-bit register pair: one v_pk_mul_f32 with the tap
// selected by op_sel (the compiler spends a v_mov on every odd tap); the product is rounded as the reference's is
template <int HI>
__device__ __forceinline__ trx_v2f pk_mul_tap(trx_v2f x, trx_v2f hpair)
{
	trx_v2f r;
	if (HI)
		asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(hpair));
	else
		asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(hpair));
	return r;
}

// Complex.h:113 norm2(): i*i + r*r
__device__ __forceinline__ float norm2(c32 v) { return v.y * v.y + v.x * v.x; }

// ------------------------------------------------------------------------------------------------
// Correlation against a GMSK training sequence without multiplications -- and still bit-identical to
// convolve_complex() (convolve_base.c:34-38, :72-85).
//
// Every GMSK correlation sequence of the reference (TSC 0..7, RACH TS0..2, dummy) is conj(+-1 rotated by k*pi/2) with
// the rotation table's fp64 phase residue left in: tap k is (+-1, e) for even k and (e, +-1) for odd k with
// |e| <= 4.6e-14 (7e-14 for the SCH sequence; bound 1e-13 checked when the tables are generated: trx_tables.unit_neg / unit_ok).  mac_cmplx() evaluates
//     y0 += x0*h0 - x1*h1;   y1 += x0*h1 + x1*h0;
// For an even tap h = (s, e): fl(s*x0) = +-x0 exactly and |fl(x1*e)| <= |x1| * 4.6e-14, which is below a quarter ulp
// of x0 -- so that fl(+-x0 - fl(x1*e)) == +-x0 -- whenever |x1| <= 2^17 |x0| (2^17 * 4.6e-14 = 6.0e-9 < 2^-26);
// likewise for y1 and for odd taps with the roles of x0 and x1 swapped.  Under that guard (TRX_UNIT_RATIO, evaluated
// once per decimated sample; it fails for about one burst in 3000, which then takes the multiplying path) each tap
// contributes +-x0 / +-x1 exactly and the reference's fl(y + t) is ONE v_pk_add_f32 with the swap / sign carried by
// op_sel / neg modifiers -- 16 VALU instructions for a normal-burst correlation instead of 64.
// The sign patterns are compile-time constants (3GPP TS 45.002 training sequences); trxhip_create*() refuses the fast
// path when they do not match what the table generator derived from the taps.
// ------------------------------------------------------------------------------------------------
#define TRX_UNIT_RATIO_LOG2 17
#define TRX_IFLAG_NO_UNIT 0x40      // bit of the kernels' `slice` argument set by the C ABI when the tables lack the unit structure
#define TRX_IFLAG_NO_SYM  0x80      // ... when the /4 decimator's taps are not bitwise symmetric (g[k] == g[15-k]): no straight-line paths
#define TRX_IFLAG_NO_FAST 0x20      // ... when the sinc LUT's absolute row sums exceed TRX_FAST_W (the FAST detector's proven margin): exact TOA search
// bit k set: the +-1 component of tap k is -1 (from the generated tables; tests/test_capi_cpu.py pins them)
#define TRX_UNIT_NEG_TSC0   0x447bull
#define TRX_UNIT_NEG_TSC1   0xc5bbull
#define TRX_UNIT_NEG_TSC2   0x7488ull
#define TRX_UNIT_NEG_TSC3   0x7709ull
#define TRX_UNIT_NEG_TSC4   0xa75cull
#define TRX_UNIT_NEG_TSC5   0xf60dull
#define TRX_UNIT_NEG_TSC6   0x4eb9ull
#define TRX_UNIT_NEG_TSC7   0xdc21ull
#define TRX_UNIT_NEG_RACH0  0xa5cc00674bull
#define TRX_UNIT_NEG_RACH1  0xfd6df886b3ull
#define TRX_UNIT_NEG_RACH2  0x4429f37d6eull
#define TRX_UNIT_NEG_DUMMY  0x1212ull
#define TRX_UNIT_NEG_SCH    0x41f73b2d69b9df04ull     /* 64 taps (trx_sch.hip); residue up to 7e-14: 2^17 * 7e-14 = 9.2e-9 < 2^-26 */

// One s_waitcnt for a block's eight LDS reads instead of the compiler's one per use: a wait is an issue slot of the wave like
// any instruction, and eight of them in front of eight multiply-adds doubled the block (round 5: - 1 % wave cycles together with
// the grouped reads of fir24x3; the same for the decimator's 16 reads and for the burst's ten global loads measured nothing)
#define TRX_FAST_ONE_WAIT() __builtin_amdgcn_s_waitcnt(0xc07f)   /* lgkmcnt(0) */

// acc += x * u for the unit tap u of parity ODD and sign NEG:  even: +-(x0, x1);  odd: h = (e, s): (-s*x1, s*x0)
template <bool ODD, bool NEG>
__device__ __forceinline__ trx_v2f unit_mac(trx_v2f acc, trx_v2f x)
{
	if (!ODD && !NEG) asm("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x));
	if (!ODD && NEG)  asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(x));
	if (ODD && !NEG)  asm("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "+v"(acc) : "v"(x));
	if (ODD && NEG)   asm("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "+v"(acc) : "v"(x));
	return acc;
}

// one tap, parity and sign folded from constants once the caller's loops are unrolled
__device__ __forceinline__ trx_v2f unit_mac_k(trx_v2f acc, trx_v2f x, unsigned long long negmask, int k)
{
	const bool odd = (k & 1) != 0, neg = ((negmask >> k) & 1ull) != 0;
	if (!odd && !neg) return unit_mac<false, false>(acc, x);
	if (!odd && neg)  return unit_mac<false, true>(acc, x);
	if (odd && !neg)  return unit_mac<true, false>(acc, x);
	return unit_mac<true, true>(acc, x);
}

// N taps through a window of 8 samples in registers (the register footprint of the multiplying loop it replaces).  A
// v_pk_add_f32 that depends on the v_pk_add_f32 in front of it costs a wait state (the compiler pads with s_nop 0), and the sum
// is ONE chain by definition (the reference's order): the read of sample k + 8 is issued straight behind the add of sample k and
// takes the pad's slot -- an instruction that has to be issued anyway.  One s_waitcnt per block: the block's samples have all
// arrived before its first add, so the reads in between need none.  Only the last block keeps its pads.
template <unsigned long long NEGMASK, int N>
struct UnitCorr {
	static __device__ __forceinline__ trx_v2f run(trx_v2f acc, const c32 *p)
	{
		static_assert(N % 8 == 0, "blocks of 8 taps");
		c32 x[8];
#pragma unroll
		for (int u = 0; u < 8; u++)
			x[u] = lds_c32(p + u);
#pragma unroll
		for (int k0 = 0; k0 < N; k0 += 8) {
			TRX_FAST_ONE_WAIT();
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u = 0; u < 8; u++) {
				acc = unit_mac_k(acc, (trx_v2f){ x[u].x, x[u].y }, NEGMASK, k0 + u);
				if (k0 + 8 < N)
					x[u] = lds_c32(p + k0 + 8 + u);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		return acc;
	}
};

// Access bursts: 40 taps over up to 80 lags.  Lane l < 40 takes the two ADJACENT lags 2l and 2l + 1: their windows share 39
// of 41 samples, so the lane reads 21 aligned sample PAIRS (ds_read_b128, conflict-free at a 16-byte lane stride) where two
// rounds of one lag per lane read 80 samples -- the correlation is bound by LDS bytes, not by its 80 additions.  Each lag
// still accumulates its taps k = 0 .. 39 in order.  p = &sig[2l + start - 39], 16-byte aligned.
template <unsigned long long NEGMASK>
struct UnitCorrPair40 {
	static __device__ __forceinline__ void run(const c32 *p, trx_v2f &a0, trx_v2f &a1)
	{
		typedef float v4f_t __attribute__((ext_vector_type(4)));
		typedef const volatile v4f_t __attribute__((address_space(3))) *lds_ptr4;
		a0 = (trx_v2f){ 0.0f, 0.0f };
		a1 = (trx_v2f){ 0.0f, 0.0f };
#pragma unroll
		for (int m0 = 0; m0 < 20; m0 += 4) {
			v4f_t x[4];
#pragma unroll
			for (int u = 0; u < 4; u++)
				x[u] = *(lds_ptr4)(p + 2 * (m0 + u));
			TRX_FAST_ONE_WAIT();
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int m = m0 + u;                                // samples 2m (a) and 2m + 1 (b) of the lane's window
				const trx_v2f a = { x[u].x, x[u].y }, b = { x[u].z, x[u].w };
				// (alternating the two lags' chains -- a1 a0 a1 a0 -- does not save the pads: a packed add between two dependent
				// packed adds does not count as their wait state, the compiler pads every second instruction either way)
				if (m >= 1)
					a1 = unit_mac_k(a1, a, NEGMASK, 2 * m - 1);
				a0 = unit_mac_k(a0, a, NEGMASK, 2 * m);
				a0 = unit_mac_k(a0, b, NEGMASK, 2 * m + 1);
				a1 = unit_mac_k(a1, b, NEGMASK, 2 * m);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		const c32 last = lds_c32(p + 40);
		a1 = unit_mac_k(a1, (trx_v2f){ last.x, last.y }, NEGMASK, 39);
	}
};

__device__ __forceinline__ void corr_unit_pair40(int slot, const c32 *p, trx_v2f &a0, trx_v2f &a1)
{
	switch (slot) {                                                 // wave-uniform: a scalar jump
	case 8: UnitCorrPair40<TRX_UNIT_NEG_RACH0>::run(p, a0, a1); break;
	case 9: UnitCorrPair40<TRX_UNIT_NEG_RACH1>::run(p, a0, a1); break;
	default: UnitCorrPair40<TRX_UNIT_NEG_RACH2>::run(p, a0, a1); break;
	}
}

// corr value for the sequence in LDS slot `slot` (0..7 TSC, 8..10 RACH, 19 dummy), p = &sig[i + start - (N-1)]
__device__ __forceinline__ trx_v2f corr_unit(int slot, const c32 *p)
{
	const trx_v2f z = { 0.0f, 0.0f };
	switch (slot) {                                                 // wave-uniform: a scalar jump
	case 0: return UnitCorr<TRX_UNIT_NEG_TSC0, 16>::run(z, p);
	case 1: return UnitCorr<TRX_UNIT_NEG_TSC1, 16>::run(z, p);
	case 2: return UnitCorr<TRX_UNIT_NEG_TSC2, 16>::run(z, p);
	case 3: return UnitCorr<TRX_UNIT_NEG_TSC3, 16>::run(z, p);
	case 4: return UnitCorr<TRX_UNIT_NEG_TSC4, 16>::run(z, p);
	case 5: return UnitCorr<TRX_UNIT_NEG_TSC5, 16>::run(z, p);
	case 6: return UnitCorr<TRX_UNIT_NEG_TSC6, 16>::run(z, p);
	case 7: return UnitCorr<TRX_UNIT_NEG_TSC7, 16>::run(z, p);
	case 8: return UnitCorr<TRX_UNIT_NEG_RACH0, 40>::run(z, p);
	case 9: return UnitCorr<TRX_UNIT_NEG_RACH1, 40>::run(z, p);
	case 10: return UnitCorr<TRX_UNIT_NEG_RACH2, 40>::run(z, p);
	default: return UnitCorr<TRX_UNIT_NEG_DUMMY, 16>::run(z, p);
	}
}

// the guard above for one (decimated) sample: true = a component is more than 2^17 times the other
__device__ __forceinline__ bool unit_unsafe(c32 v)
{
	const float lo = fminf(fabsf(v.x), fabsf(v.y)), hi = fmaxf(fabsf(v.x), fabsf(v.y));
	return ldexpf(lo, TRX_UNIT_RATIO_LOG2) < hi;
}

// ------------------------------------------------------------------------------------------------
// interpolatePoint() for one candidate position per lane (sigProcLib.cpp:1100-1118)
//   cz    = zero-padded correlation in LDS: cz[i] = corr[i] for 0 <= i < size-1, 0 elsewhere in
//           [-TRX_CZ_PAD, size + TRX_CZ_PAD).  The reference sums i in [max(0,fl-10), min(size-1,fl+11)):
//           note the last sample (size-1) is never used (":1105 end = size-1; i < end"), hence zeroed.
//   ix512 = position in 1/512 symbol units (multiples of 1/512 are all peakDetect() ever asks for)
//   sincv = swizzled LDS table, sincv[swz(q)] = sinc(M_PI_F * q/512), 0 for q >= 4096
// Of the 21 taps only i = fl-7 .. fl+8 can be non-zero (|i - ix| < 8, the LUT is 0 beyond 8*pi);
// dropping the others only drops additions of +-0.  q = |i*512 - ix512| is affine in the tap index on
// either side of the peak, so the LUT address is one per-lane base plus a compile-time offset.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ c32 interp_taps(const c32 *c, const float *sa, const float *sb)
{
	// pin tap 0's LDS address in a register: all 16 reads then use non-negative immediate offsets (otherwise the
	// compiler rebases on the peak and spends a VALU subtract per tap on the negative side)
	typedef const volatile trx_v2f __attribute__((address_space(3))) *lds_ptr;
	lds_ptr cp = (lds_ptr)c;
	asm("" : "+v"(cp));
	c32 p = make_float2(0.0f, 0.0f);
#pragma unroll
	for (int u = 0; u < 8; u++) {                    // i = fl-7 .. fl   (k = 7 .. 0)
		const trx_v2f v = cp[u];
		const float w = sa[512 * (7 - u)];
		p.x += v.x * w;
		p.y += v.y * w;
	}
#pragma unroll
	for (int u = 0; u < 8; u++) {                    // i = fl+1 .. fl+8 (k = 0 .. 7)
		const trx_v2f v = cp[8 + u];
		const float w = sb[512 * u];
		p.x += v.x * w;
		p.y += v.y * w;
	}
	return p;
}

// the same with the 16 weights handed over in registers (tap order: i = fl-7 .. fl+8), for positions whose weights the caller
// keeps per lane (round A of the speculative bisection: functions of the lane only)
__device__ __forceinline__ c32 interp_taps_w(const c32 *c, const float4 &w0, const float4 &w1, const float4 &w2, const float4 &w3)
{
	typedef const volatile trx_v2f __attribute__((address_space(3))) *lds_ptr;
	lds_ptr cp = (lds_ptr)c;
	asm("" : "+v"(cp));
	const float w[16] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w };
	c32 p = make_float2(0.0f, 0.0f);
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const trx_v2f v = cp[u];
		p.x += v.x * w[u];
		p.y += v.y * w[u];
	}
	return p;
}

// LUT bases of a fractional position f (= ix512 & 511): taps i <= fl use q = 512*(fl-i) + f,
// taps i > fl use q = 512*(i-fl-1) + (512 - f)
__device__ __forceinline__ int sinc_base_lo(int f) { return trx_sincv_swz(f); }
__device__ __forceinline__ int sinc_base_hi(int f) { return (f == 0) ? 512 : trx_sincv_swz(512 - f); }

__device__ __forceinline__ c32 interp_point(const c32 *cz, int ix512, const float *sincv)
{
	const int fl = ix512 >> 9;                       // floor(ix)
	const int f = ix512 & 511;                       // fractional part * 512
	return interp_taps(cz + (fl - 7), sincv + sinc_base_lo(f), sincv + sinc_base_hi(f));
}

// Lane constants of the speculative bisection (functions of the lane id only): computed once per kernel,
// they cost 5 VGPRs instead of ~40 VALU instructions per burst.
struct PeakConst {
	int flA;        // round A: floor((offset of this lane's position) / 512)
	int loA, hiA;   // round A: sinc LUT bases (the fraction depends on the lane only: E is a multiple of 512)
	int offB;       // round B: position offset of this lane (node early/late, or one of the 16 final positions)
	int ratio_off;  // computePeakRatio: offset of this lane's term, lane & 7 -> -2, +2, -3, +3, -4, +4, -5, +5
};
__device__ __forceinline__ int node_offset(int n, int inc0);
__device__ __forceinline__ PeakConst peak_const(int lane)
{
	PeakConst pc;
	const int offA = node_offset(lane >> 1, 256) + ((lane & 1) ? 1024 : 0);
	pc.flA = offA >> 9;
	pc.loA = sinc_base_lo(offA & 511);
	pc.hiA = sinc_base_hi(offA & 511);
	pc.offB = (lane < 32) ? node_offset(lane >> 1, 8) + ((lane & 1) ? 1024 : 0) : (2 * (lane & 15) - 15) + 512;
	pc.ratio_off = ((lane & 1) ? 1 : -1) * (2 + ((lane & 7) >> 1));
	return pc;
}

// earlyIndex offset (1/512 units) of heap node n of a bisection subtree whose first step is `inc0`:
// node n = (1<<L) + p - 1 at level L with path bits p (MSB first, 1 = "+incr"):
//   off = sum_{j<L} (+-)(inc0 >> j) = ((4*p*inc0) >> L) - (2*inc0 - ((2*inc0) >> L))
__device__ __forceinline__ int node_offset(int n, int inc0)
{
	const int L = 31 - __clz(n + 1);
	const int p = n + 1 - (1 << L);
	return ((4 * p * inc0) >> L) - (2 * inc0 - ((2 * inc0) >> L));
}

// One round of the speculative bisection: lanes 2n / 2n+1 hold |interp(early)|^2 / |interp(late)|^2 of
// heap node n.  Each even lane compares with its odd neighbour (one DPP move); the two v_cmp results are
// ballots the scalar unit walks: no further vector work.  Returns the accumulated step; sets `tie` at
// ":1170 else break".
template <int LEVELS>
__device__ __forceinline__ int walk_tree(float nv, int inc0, bool &tie)
{
	const float other = dpp<DPP_QUAD_XOR1, 0xf>(nv);               // even lane <- late, odd lane <- early
	const unsigned long long lt = __ballot(nv < other);            // even bits: early < late
	const unsigned long long gt = __ballot(nv > other);            // even bits: early > late
	const unsigned long long eq = ~(lt | gt);                      // neither (":1170 else break")
	// Fast walk: with no tie anywhere in the tree (exact equality of two interpolated powers: practically never) the
	// path is its decision bits p = b0 b1 ... (MSB first), node (L, p) sits at ballot bit 2 * ((1 << L) - 1 + p), and
	//     off = sum_L (b_L ? +s_L : -s_L) = 2 * s_last * p - (2 * inc0 - s_last),  s_L = inc0 >> L.
	// Three scalar instructions per level: bit position, s_bitcmp1 -> SCC, p = 2p + SCC.
	constexpr unsigned long long NODES = (LEVELS == 5) ? 0x1555555555555555ull : 0x15555555ull;   // even bits of 31 / 15 nodes
	if (!tie && (eq & NODES) == 0ull) {
		unsigned p = 0, pos;
#pragma unroll
		for (int Lw = 0; Lw < LEVELS; Lw++)
			asm volatile("s_lshl1_add_u32 %1, %0, %3\n\t"
				     "s_bitcmp1_b64 %2, %1\n\t"
				     "s_addc_u32 %0, %0, %0"
				     : "+s"(p), "=&s"(pos) : "s"(lt), "n"(2 * ((1 << Lw) - 1)) : "scc");
		const int s_last = inc0 >> (LEVELS - 1);
		return 2 * s_last * (int)p - (2 * inc0 - s_last);
	}
	// careful walk (a tie on the path, or carried in): ~10 SALU ops per level.  After a tie the offset stops changing,
	// as in the reference.
	unsigned node = 0, stop = tie ? 1u : 0u;
	int off = 0;
#pragma unroll
	for (int Lw = 0; Lw < LEVELS; Lw++) {
		const unsigned bit = 2u * node;
		const unsigned l = (unsigned)(lt >> bit) & 1u;
		stop |= (unsigned)(eq >> bit) & 1u;
		const int step = inc0 >> Lw;
		const int delta = l ? step : -step;
		off += stop ? 0 : delta;
		node = 2u * node + 1u + l;
	}
	tie = stop != 0u;
	return off;
}

// ------------------------------------------------------------------------------------------------
// peakDetect() (sigProcLib.cpp:1141-1186) with the early/late bisection expanded across lanes.
// All lanes return the same (toa512, value).
// ------------------------------------------------------------------------------------------------
//   wa4 : optional (4-SPS fused kernel) LDS table of round A's weights, float4 q of lane l at wa4[64 q + l] (the weights of
//         round A depend on the lane only): four conflict-free 16-byte reads instead of sixteen 4-byte gathers
__device__ __forceinline__ void peak_detect_spec(const c32 *cz, int max_idx, const float *sincv, const PeakConst &pc,
						  int lane, int *toa512_out, c32 *val_out, const float4 *wa4 = nullptr)
{
	int E = (max_idx - 1) * 512;                     // earlyIndex * 512
	bool tie = false;

	// ---- round A: levels 0..4 (incr = 1/2 .. 1/32): heap node n on lanes 2n (early) and 2n+1 (late)
	{
		// ix = E + offA with E a multiple of 512: floor and fraction come from the lane constants
		const c32 *const ca = cz + ((max_idx - 1) + pc.flA - 7);
		const float nv = wa4 ? norm2(interp_taps_w(ca, wa4[lane], wa4[64 + lane], wa4[128 + lane], wa4[192 + lane]))
				     : norm2(interp_taps(ca, sincv + pc.loA, sincv + pc.hiA));
		// (lanes 62,63 evaluate a harmless extra node)
		E += walk_tree<5>(nv, 256, tie);
	}
	int final_ix;
	c32 val;
	if (!tie) {
		// ---- round B: levels 5..8 (incr = 1/64 .. 1/512) on lanes 0..29, the 16 possible final
		// positions (earlyIndex + 1) on lanes 32..47
		const c32 pv = interp_point(cz, E + pc.offB, sincv);
		const float nv = norm2(pv);
		const int offB = walk_tree<4>(nv, 8, tie);
		E += offB;
		final_ix = E + 512;
		const int src = 32 + ((offB + 15) >> 1);     // lane that evaluated this final position
		val.x = lane_val(pv.x, src);
		val.y = lane_val(pv.y, src);
	}
	if (tie) {                                        // rare: equal early/late power -> loop left early
		final_ix = E + 512;
		val = interp_point(cz, final_ix, sincv);
	}
	*toa512_out = final_ix;
	*val_out = val;
}

// ------------------------------------------------------------------------------------------------
// The FAST detector of the fused 4-SPS kernels (round 5): the two interpolation rounds of the TOA bisection with FMA --
// 16 v_pk_fma_f32 per round instead of 16 v_pk_mul_f32 + 16 v_pk_add_f32 -- and every early / late decision certified by a
// PROVEN margin; a burst with an uncertified decision on its path re-runs peak_detect_spec() in the reference's operand order.
// rc and TOA are therefore identical to the reference's by construction; the interpolated peak value (-> amp, C/I) differs
// from the reference's by the rounding of one 16-term sum (TRXHIP_FAST_AMP_RTOL, include/trxhip.h).
//
// The bound.  u = 2^-24.  Per component, the reference's sum s = fl(s + fl(c w)) is within gamma_16 * sum |c_u| |w_u| of the true
// sum (a term passes through at most 16 roundings); the FMA sum -- two chains of eight, s = fl(s + c w), joined by one addition --
// within gamma_9 * sum |c_u| |w_u|;  gamma_n < 1.0001 n u.  Every correlation sample the interpolation reads has
// |c_u|^2 <= m, the arg-max power (the zero pads included), and sum_u |w_u| <= W = 2.6 over all 512 fractional positions
// (TRX_FAST_W; the sinc LUT's maximum is 2.573, checked when the context is created).  So per component
//     |p_fma - p_ref| <= e1 = 25.003 u * 2.6 * sqrt(m) < 3.88e-6 sqrt(m),
// and for the powers N = |p|^2:
//     |N_fma - N_ref| <= |p_fma - p_ref| (|p_fma| + |p_ref|) <= sqrt(2) e1 (2 sqrt(N_fma) + sqrt(2) e1)
//                     <= sqrt(2) * 3.88e-6 * (m + N_fma) + 3.1e-11 m                 (2 sqrt(m N) <= m + N).
// norm2() rounds three times: the computed nv is within 2.0001 u N of N on both sides.  Together
//     |nv_ref - nv_fma| <= 5.75e-6 * (m + nv_fma)   and the kernel uses   r = TRX_FAST_KAPPA * (nv_fma + m), KAPPA = 6e-6
// (the 2.5e-7 (m + nv) of slack covers the roundings of r, nv - r, nv + r themselves): nv_ref lies in [nv - r, nv + r].
// "early < late" is certain when  nv_E + r_E < nv_L - r_L,  "early > late" when  nv_L + r_L < nv_E - r_E  -- one compare
// per lane of its own upper bound with its neighbour's lower bound gives both (even lanes: the first, odd lanes: the second).
// ------------------------------------------------------------------------------------------------
#define TRX_FAST_W      2.6f
#define TRX_FAST_KAPPA  6e-6f
// [0]: bursts whose TOA search was re-run in the reference's operand order since the last reset (one atomic per such burst,
// in the cold path; read through trxhip_fast_stats()).  One copy per translation unit; only trx_kernel4.hip's is ever written.
static __device__ unsigned long long g_trx_fast_stats[4];

__device__ __forceinline__ trx_v2f pk_fma_w(trx_v2f x, float w, trx_v2f acc)
{
	// acc += x * w, w broadcast from the low half of its register pair (the high half is never read)
	return __builtin_elementwise_fma(x, (trx_v2f){ w, w }, acc);
}

// Two chains (even / odd taps) joined by one addition: a dependent v_pk_fma_f32 needs a wait state behind its predecessor,
// two independent ones do not -- and a term then passes through at most 9 roundings, not 16.  The sixteen taps run as two
// blocks of eight reads + eight FMAs (the empty volatile asm between them pins the order: with all sixteen samples and
// weights in flight the kernel needs 48 registers here and spills elsewhere).
__device__ __forceinline__ c32 interp_taps_fma(const c32 *c, const float *sa, const float *sb)
{
	typedef const volatile trx_v2f __attribute__((address_space(3))) *lds_ptr;
	lds_ptr cp = (lds_ptr)c;
	asm("" : "+v"(cp));
	trx_v2f p0 = { 0.0f, 0.0f }, p1 = { 0.0f, 0.0f };
#pragma unroll
	for (int h = 0; h < 2; h++) {
		trx_v2f x[8];
		float w[8];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			x[u] = cp[8 * h + u];
			w[u] = h ? sb[512 * u] : sa[512 * (7 - u)];
		}
		TRX_FAST_ONE_WAIT();
#pragma unroll
		for (int u = 0; u < 8; u += 2) {
			p0 = pk_fma_w(x[u], w[u], p0);
			p1 = pk_fma_w(x[u + 1], w[u + 1], p1);
		}
		asm volatile("" : "+v"(p0), "+v"(p1));
	}
	const trx_v2f p = p0 + p1;
	return make_float2(p.x, p.y);
}

// the same with the weights in an LDS table of float4: quad q of this lane at wq[64 q] (round A: functions of the lane only)
__device__ __forceinline__ c32 interp_taps_w_fma(const c32 *c, const float4 *wq)
{
	typedef const volatile trx_v2f __attribute__((address_space(3))) *lds_ptr;
	lds_ptr cp = (lds_ptr)c;
	asm("" : "+v"(cp));
	trx_v2f p0 = { 0.0f, 0.0f }, p1 = { 0.0f, 0.0f };
#pragma unroll
	for (int h = 0; h < 2; h++) {
		trx_v2f x[8];
		const float4 qa = wq[64 * (2 * h)], qb = wq[64 * (2 * h + 1)];
#pragma unroll
		for (int u = 0; u < 8; u++)
			x[u] = cp[8 * h + u];
		TRX_FAST_ONE_WAIT();
#pragma unroll
		for (int u = 0; u < 8; u += 2) {
			const float4 q = (u & 4) ? qb : qa;
			const trx_v2f hp = (u & 2) ? (trx_v2f){ q.z, q.w } : (trx_v2f){ q.x, q.y };
			p0 = pk_fma_tap<0>(x[u], hp, p0);
			p1 = pk_fma_tap<1>(x[u + 1], hp, p1);
		}
		asm volatile("" : "+v"(p0), "+v"(p1));
	}
	const trx_v2f p = p0 + p1;
	return make_float2(p.x, p.y);
}

// walk_tree() on certified comparisons: lanes 2n / 2n+1 hold nv of heap node n's early / late position, km = KAPPA * m.
// Returns the accumulated step; sets `unsure` when a node ON THE PATH has no certified decision (an exact tie included).
template <int LEVELS>
__device__ __forceinline__ int walk_tree_fast(float nv, float km, int inc0, bool &unsure)
{
	const float r = fmaf(nv, TRX_FAST_KAPPA, km);
	const float lo = nv - r, hi = nv + r;
	const float other_lo = dpp<DPP_QUAD_XOR1, 0xf>(lo);
	const unsigned long long c = __ballot(hi < other_lo);          // even bits: early < late for sure; odd bits: early > late for sure
	constexpr unsigned long long NODES = (LEVELS == 5) ? 0x1555555555555555ull : 0x15555555ull;
	if (__builtin_expect(((c | (c >> 1)) & NODES) == NODES, 1)) {  // every node of the tree decided: the three-instruction walk
		unsigned p = 0, pos;
#pragma unroll
		for (int Lw = 0; Lw < LEVELS; Lw++)
			asm volatile("s_lshl1_add_u32 %1, %0, %3\n\t"
				     "s_bitcmp1_b64 %2, %1\n\t"
				     "s_addc_u32 %0, %0, %0"
				     : "+s"(p), "=&s"(pos) : "s"(c), "n"(2 * ((1 << Lw) - 1)) : "scc");
		const int s_last = inc0 >> (LEVELS - 1);
		return 2 * s_last * (int)p - (2 * inc0 - s_last);
	}
	// some node undecided: only those on the path matter
	unsigned node = 0;
	int off = 0;
#pragma unroll
	for (int Lw = 0; Lw < LEVELS; Lw++) {
		const unsigned bit = 2u * node;
		const unsigned l = (unsigned)(c >> bit) & 1u, g = (unsigned)(c >> (bit + 1u)) & 1u;
		if (!(l | g))
			unsure = true;
		const int step = inc0 >> Lw;
		off += l ? step : -step;
		node = 2u * node + 1u + l;
	}
	return off;
}

// peak_detect_spec() with FMA sums; *unsure: a decision on the path is not certified (the caller re-runs the exact one)
//   wa4: round A's weight table (see peak_detect_spec); every caller of the FAST detector has one
__device__ __forceinline__ void peak_detect_fast(const c32 *cz, int max_idx, float m, const float *sincv, const PeakConst &pc,
						  int lane, int *toa512_out, c32 *val_out, const float4 *wa4, bool *unsure_out)
{
	int E = (max_idx - 1) * 512;
	bool unsure = false;
	const float km = TRX_FAST_KAPPA * m;
	{
		const c32 *const ca = cz + ((max_idx - 1) + pc.flA - 7);
		const float nv = norm2(interp_taps_w_fma(ca, wa4 + lane));
		E += walk_tree_fast<5>(nv, km, 256, unsure);
	}
	c32 val = make_float2(0.0f, 0.0f);
	int final_ix = E + 512;
	if (!unsure) {
		const int ixb = E + pc.offB;
		c32 pv = make_float2(0.0f, 0.0f);
		if (lane < 30 || (lane >= 32 && lane < 48))                      // 15 nodes x {early, late} + 16 final positions: 18 lanes idle
			pv = interp_taps_fma(cz + ((ixb >> 9) - 7), sincv + sinc_base_lo(ixb & 511), sincv + sinc_base_hi(ixb & 511));
		const int offB = walk_tree_fast<4>(norm2(pv), km, 8, unsure);
		E += offB;
		final_ix = E + 512;
		const int src = 32 + ((offB + 15) >> 1);
		val.x = lane_val(pv.x, src);
		val.y = lane_val(pv.y, src);
	}
	*toa512_out = final_ix;
	*val_out = val;
	*unsure_out = unsure;
}

// ------------------------------------------------------------------------------------------------
// detectBurst() after fastPeakDetect (sigProcLib.cpp:1683-1708): edge gate, computePeakRatio gate, peakDetect,
// computeCI, amp and toa.  One wave; `bidx` is the wave-uniform index of the first strict maximum of |corr|^2.
//   cz     : correlation with TRX_CZ_PAD zeros either side; only cz[bidx-12 .. bidx+12] is read, so a caller
//            with a long correlation (SCH buffer search) may pass a 25-sample window, biased so that it is
//            indexed by the absolute position (the ":1105" zeroing of cz[len-1] is range-checked for that)
//   sig    : what was correlated (computeCI reads N samples of it), sig_len its length
//   hdr    : {gain.re, gain.im, ginv.re, ginv.im, ci_den, toa, n, 1/ci_den}
// ------------------------------------------------------------------------------------------------
//   on_toa : called with the refined position (1/512 symbol units, before "- sync->toa" and "- head") as soon as peakDetect
//            has it -- the 4-SPS kernel starts fetching what its demodulator needs for that TOA behind computeCI
struct NoToaHook { __device__ __forceinline__ void operator()(int) const {} };
template <bool FAST, typename Hook>
__device__ __forceinline__ int detect_tail_h(const c32 *sig, int sig_len, c32 *cz, const float *hdr, int N, float thresh,
					      int start, int len, int bidx, const float *sincv, const PeakConst &pc, int lane,
					      float *toa_out, c32 *amp_out, float *ci_out, Hook on_toa, const float4 *wa4, int slice DIAG_ARG WI_ARG)
{
	if (!WI_SKIP && ((bidx < 3) || (bidx > len - 3)))   // :1683
		return 0;
	wave_sync();
	const c32 amp0 = cz[bidx];

	if (ABL(9)) { *toa_out = (float)bidx; *amp_out = amp0; *ci_out = 0.0f; return 1; }
	DIAG_MARK(4);
	// ---- computePeakRatio (:1541-1571): terms in the reference's order; out-of-range terms read the
	// zero pads (adding +0 is exact), their count is arithmetic
	const float pwr = WI_SKIP ? 0.0f : norm2(cz[bidx + pc.ratio_off]);   // (read before cz[len - 1] is zeroed below)
	auto ratio_gate = [&]() -> bool {
		// lane k (mod 8) squares the k-th term of the reference's loop (peak-2, peak+2, peak-3, ... peak+5; out-of-range
		// ones read the zero pads), a serial DPP scan adds them left to right: lane 7 holds the reference's avg
		auto ordered_avg = [&]() {
			float acc = pwr;
#pragma unroll
			for (int i = 1; i < 8; i++)
				asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(pwr));
			return lane_val(acc, 7);
		};
		float avg;
		if (FAST) {
			// FAST: the eight terms tree-summed inside their group of eight lanes (3 DPP steps instead of 7): within
			// (7 + 3) u = 6e-7 of the ordered sum, which the estimate's margin below absorbs; the exactly rounded path
			// behind the margin re-sums them in the reference's order
			float acc = pwr;
			asm volatile(TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
				     TRX_DPP_STEP("v_add_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
				     TRX_DPP_STEP("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
				     : "+v"(acc));
			avg = lane_val(acc, 0);
		} else {
			avg = ordered_avg();
		}
		// number of in-range terms (:1555-1562).  The edge gate above left 3 <= bidx <= len - 3, so the four terms at
		// distance 2 and 3 always exist except peak + 3 == len: 7 or 8 >= 5 ("num < 5: return 0" can never fire here)
		// = 3 + (bidx + 3 < len) + (bidx >= 4) + (bidx + 4 < len) + (bidx >= 5) + (bidx + 5 < len), as scalar min / add:
		// terms at distance 2..5 exist on the left while bidx - d >= 0, on the right while bidx + d < len
		const int nl = bidx - 1, nr = len - 2 - bidx;
		const int num = (nl < 4 ? nl : 4) + (nr < 4 ? nr : 4);
		// The gate "|amp| / (sqrtf(avg/num) + 1e-5) < thresh" is a decision, so it must round as the reference
		// does -- but two IEEE divisions, two correctly rounded square roots and an fp64 add cost ~45 VALU ops.
		// A 1-ulp-per-op estimate (total error < 1e-6) decides every burst whose ratio is not within 4e-6 of
		// the threshold; only those (about one in 1e5) take the exactly rounded path.
		// Compared as squares: |amp|^2 against (thresh * rms)^2 -- one transcendental (the sqrt of the mean) instead of four;
		// 1 / num is a scalar select (num is 7 or 8 after the edge gate), the margins are the squares of the linear ones.
		const float amp2 = norm2(amp0);
		const float rnum = (num == 8) ? 0.125f : (num == 7) ? (1.0f / 7.0f) : (num == 6) ? (1.0f / 6.0f) : 0.2f;
		const float rms_e = __builtin_amdgcn_sqrtf(avg * rnum) + 0.00001f;
		const float t_e = thresh * rms_e;
		const float t2 = t_e * t_e;
		const float gm = FAST ? 1.2e-5f : 8e-6f;                   // (FAST: + the tree sum's 6e-7, with room)
		if (amp2 < t2 * (1.0f - gm))
			return false;
		if (!(amp2 > t2 * (1.0f + gm))) {
			if (FAST)
				avg = ordered_avg();
			const float rms = (float)((double)sqrtf(avg / (float)num) + 0.00001);
			const float ratio = sqrtf(amp2) / rms;
			if (ratio < thresh)
				return false;
		}
		return true;
	};
	// (Round 5 measured the other order for the 4-SPS kernel -- TOA search first, so that its hook's loads of the demodulator's
	// low-edge tap rows have the gate's ~250 cycles more to arrive; the wait for those rows is 1.8 % of the kernel -- and did
	// not keep it: -0.8 %, the extra search on the slots that fail the gate costs more; profiles/r05_ab_runs.txt.)
	if (!WI_SKIP && !ratio_gate())
		return 0;

	DIAG_MARK(5);
	// ---- peakDetect (:1695): refined TOA (multiple of 1/512) and interpolated correlation value
	int toa512;
	c32 xcorr;
	// interpolatePoint() never reads the last correlation sample (:1105, :1109): zero it in the padded copy
	if (lane == 0 && len - 1 - bidx <= TRX_CZ_PAD)       // (farther from the peak nothing reads it)
		cz[len - 1] = make_float2(0.0f, 0.0f);
	wave_sync();
	if (ABL(1)) { toa512 = bidx * 512; xcorr = amp0; }
	else if (FAST && !(slice & TRX_IFLAG_NO_FAST)) {
		// FMA sums, decisions certified against m = |corr[bidx]|^2 (no correlation sample is larger); the exact search
		// only for the bursts with an uncertified decision on their path
		bool unsure;
		peak_detect_fast(cz, bidx, norm2(amp0), sincv, pc, lane, &toa512, &xcorr, wa4, &unsure);
		if (unsure) {
			if (lane == 0)
				atomicAdd(&g_trx_fast_stats[0], 1ull);
			peak_detect_spec(cz, bidx, sincv, pc, lane, &toa512, &xcorr, wa4);
		}
	} else
		peak_detect_spec(cz, bidx, sincv, pc, lane, &toa512, &xcorr, wa4);
	toa512 = uni(toa512);
	xcorr.x = unif(xcorr.x);
	xcorr.y = unif(xcorr.y);
	const float toa = (float)toa512 * (1.0f / 512.0f);   // exact
	on_toa(toa512);

	DIAG_MARK(6);
	// ---- computeCI (:1608-1639)
	float ci = 0.0f;
	{
		// roundf(toa): toa is k/512 -> round half away from zero on integers
		const int rt = (toa512 >= 0) ? ((toa512 + 256) >> 9) : -((-toa512 + 256) >> 9);
		const int ps = start + 1 - N + rt;
		if (ps >= 0 && ps + N <= sig_len && !ABL(7) && !WI_SKIP) {
			// S = sum_i |sig[ps+i]|^2 in index order: lane i squares one sample, the sum walks the lanes
			const float pw = norm2(sig[ps + (lane < N ? lane : 0)]);
			// serial scan along the lanes: after step s lane k holds pw[k-s] + ... + pw[k] added left to right, so
			// lane N-1 ends with the reference's S (one DPP add per term instead of a readlane + add)
			float acc = pw;
#define TRX_SCAN_STEP asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(pw))
			float S;
			if (FAST) {
				// FAST: C/I is a tolerance quantity already (C is an FMA sum), so S is tree-summed: 4 DPP steps over the
				// first row of lanes (N = 16), the wave tree over the first N lanes otherwise (access bursts: 6 steps for 39)
				if (N == 16) {
					S = lane_val(row_sum(pw), 0);
				} else {
					S = wave_sum(lane < N ? pw : 0.0f);
				}
			} else {
				if (N == 16) {                                   // normal bursts: straight-line
#pragma unroll
					for (int i = 1; i < 16; i++)
						TRX_SCAN_STEP;
				} else {
					for (int i = 1; i < N; i++)
						TRX_SCAN_STEP;
				}
				S = lane_val(acc, N - 1);
			}
#undef TRX_SCAN_STEP
			// S - C cancels (by a factor C/I), so S and C themselves must round as the reference's do; only the last
			// quotient and the log may be approximate (2 ulp, hardware log2).  a / b with a correctly rounded
			// reciprocal y of b: q = a*y, r = fma(-q, b, a), q' = fma(r, y, q) is the correctly rounded quotient
			// (Markstein) -- three instructions instead of an IEEE division sequence.
			auto div_y = [](float a, float b, float y) { const float q = a * y; return fmaf(fmaf(-q, b, a), y, q); };
			float C;
			if (FAST) {                                   // tolerance quantities: one rounding each instead of the exact quotients
				S *= (N == 40) ? 0.025f : (N == 16) ? 0.0625f : 0.015625f;
				C = norm2(xcorr) * hdr[7];
			} else {
				if (N == 40) S = div_y(S, 40.0f, 0.025f);        // S /= N (:1631); 1/16 and 1/64 are exact
				else S *= (N == 16) ? 0.0625f : 0.015625f;
				C = div_y(norm2(xcorr), hdr[4], hdr[7]);         // / ((N-1)*|gain|) (:1633), table: ci_den and RN(1/ci_den)
			}
			ci = 3.0103f * __log2f(C * __builtin_amdgcn_rcpf(S - C));
		}
	}

	const c32 a = cmul(xcorr, make_float2(hdr[2], hdr[3]));  // xcorr / sync->gain  (:1701)
	*amp_out = make_float2(unif(a.x), unif(a.y));            // wave-uniform: lives in scalar registers from here on
	*toa_out = unif(toa - hdr[5]);                           // :1704
	*ci_out = ci;
	DIAG_MARK(7);
	return 1;
}

__device__ __forceinline__ int detect_tail(const c32 *sig, int sig_len, c32 *cz, const float *hdr, int N, float thresh,
					    int start, int len, int bidx, const float *sincv, const PeakConst &pc, int lane,
					    float *toa_out, c32 *amp_out, float *ci_out, int slice DIAG_ARG)
{
	WI_LOCAL;
	return detect_tail_h<false>(sig, sig_len, cz, hdr, N, thresh, start, len, bidx, sincv, pc, lane, toa_out, amp_out, ci_out,
				    NoToaHook(), nullptr, slice DIAG_PASS WI_PASS);
}

// ------------------------------------------------------------------------------------------------
// detectBurst() on the 1-SPS signal `sig` (sigProcLib.cpp:1649-1709); correlation kept in LDS (cz).
//   PADDED: sig is readable (zero) over the whole correlation window, no range checks (4 SPS: dec[])
//   taps  : LDS, wave-uniform -> broadcast reads; hdr: {gain.re, gain.im, ginv.re, ginv.im, ci_den, toa}
// Returns rc (1 / 0); on 1 fills toa (symbols, before "- head"), amp, ci.  Wave-uniform.
// ------------------------------------------------------------------------------------------------
//   NARROW: sig[] and cz[] are the TRX_DEC_NARROW / TRX_CORR_NARROW buffers: a window that does not fit them
//           (max_toa > 64) is correlated without storing it and only the 25 values around the peak are
//           recomputed for the tail, exactly as the SCH buffer search does (trx_sch.hip)
//   unit_slot >= 0: the sequence is the GMSK one of that LDS slot and every sample the window reads passed
//           unit_unsafe(): correlate with corr_unit() (additions only, same bits); < 0: multiply as written
//   FAST  : the TOA search with FMA sums and certified decisions (peak_detect_fast); amp and C/I are then tolerance quantities
template <bool PADDED, bool NARROW, bool FAST, typename Hook>
__device__ __forceinline__ int detect_burst_h(const c32 *sig, int sig_len, c32 *cz, const c32 *taps, const float *hdr,
					       int N, float thresh, int start, int len, const float *sincv, const PeakConst &pc, int lane,
					       float *toa_out, c32 *amp_out, float *ci_out, Hook on_toa, const float4 *wa4, int slice, int unit_slot DIAG_ARG WI_ARG)
{
#ifdef TRX_WHATIF_PAIR
	if (wi.skip)                                     // the neighbour's pass "already" correlated and gated this burst
		return detect_tail_h<FAST>(sig, sig_len, cz, hdr, N, thresh, start, len, wi.bidx, sincv, pc, lane, toa_out, amp_out, ci_out,
					   on_toa, wa4, slice DIAG_PASS WI_PASS);
#endif
	const bool wide = NARROW && (len > TRX_CORR_NARROW || start + len > TRX_DEC_NARROW);
	// corr[i] with range-checked reads, taps in order (cold: wide windows only)
	auto corr_at = [&](int i) {
		float yr = 0.0f, yi = 0.0f;
		const int base = i + start - (N - 1);
		for (int k = 0; k < N; k++) {
			const int j = base + k;
			const c32 x = (j >= 0 && j < sig_len) ? sig[j] : make_float2(0.0f, 0.0f);
			const c32 h = taps[k];
			yr += x.x * h.x - x.y * h.y;
			yi += x.x * h.y + x.y * h.x;
		}
		return make_float2(yr, yi);
	};

	// ---- correlate: corr[i] = sum_k SIG(i + start - (N-1) + k) * seq[k]   (:1674, convolve_base.c:72-85)
	// N is 16 (TSC/EDGE) or 40 (RACH): tap loop unrolled by 8 so the LDS reads pipeline
	float best = 0.0f;                               // fastPeakDetect state, fused into the same pass
	int bidx = -1;
	bool pair = false;                               // two adjacent lags per lane (access bursts)
	if (wide) {
		for (int i = lane; i < len; i += WAVE) {
			const float v = norm2(corr_at(i));
			if (v > best) { best = v; bidx = i; }
		}
	} else if (PADDED && unit_slot >= 8 && unit_slot <= 10 && N == 40 && len <= 80 && ((start - 39) & 1) == 0) {
		// access bursts: lags 2 lane and 2 lane + 1 on lanes 0 .. 39 (corr_unit_pair40)
		pair = true;
		const int l2 = lane < 40 ? 2 * lane : 78;
		trx_v2f a0, a1;
		a0 = (trx_v2f){ 0.0f, 0.0f };
		a1 = a0;
		if (lane < 40)
			corr_unit_pair40(unit_slot, sig + (l2 + start - 39), a0, a1);
		const bool in0 = lane < 40 && l2 < len, in1 = lane < 40 && l2 + 1 < len;
		// lags >= len are not part of the correlation: zeros (cz[len ..] is the right zero pad)
		const float4 y = make_float4(in0 ? a0.x : 0.0f, in0 ? a0.y : 0.0f, in1 ? a1.x : 0.0f, in1 ? a1.y : 0.0f);
		if (lane < 40)
			*reinterpret_cast<float4 *>(cz + l2) = y;
		if (lane >= 40 && lane < 40 + TRX_CZ_PAD)
			cz[80 + (lane - 40)] = make_float2(0.0f, 0.0f);          // with the zeros above: cz[len .. 91], the right zero pad
		// fastPeakDetect: first strict maximum -- inside the lane lag 2l before 2l + 1, across lanes the lowest lane (below)
		const float v0 = norm2(make_float2(y.x, y.y)), v1 = norm2(make_float2(y.z, y.w));
		best = v0; bidx = l2;
		if (v1 > v0) { best = v1; bidx = l2 + 1; }
		if (!(best > 0.0f)) bidx = -1;
	} else if (PADDED && unit_slot >= 0 && len + TRX_CZ_PAD <= WAVE && start + WAVE <= sig_len + 4) {
		// one round (normal bursts: len <= 49): lane = lag, and the twelve lanes behind the window write the right zero pad in
		// the SAME store (an LDS store costs three reads); what they correlated -- real samples further on -- is discarded
		const bool in = lane < len;
		trx_v2f acc = { 0.0f, 0.0f };
		if (in)                                                         // (idle lanes neither read nor add: LDS time and power)
			acc = corr_unit(unit_slot, sig + (lane + start - (N - 1)));
		const c32 y = make_float2(acc.x, acc.y);
		if (lane < len + TRX_CZ_PAD)
			cz[lane] = y;
		const float v = norm2(y);
		if (v > best) { best = v; bidx = lane; }
	} else if (PADDED && unit_slot >= 0) {
		for (int i = lane; i < len; i += WAVE) {
			const trx_v2f acc = corr_unit(unit_slot, sig + (i + start - (N - 1)));
			const c32 y = make_float2(acc.x, acc.y);
			cz[i] = y;
			const float v = norm2(y);
			if (v > best) { best = v; bidx = i; }
		}
		if (lane < TRX_CZ_PAD)
			cz[len + lane] = make_float2(0.0f, 0.0f);
	} else {
		for (int i = lane; i < len; i += WAVE) {
			// (yr, yi) += (xr*hr - xi*hi, xr*hi + xi*hr) as four packed ops per tap: two v_pk_mul, one v_pk_add with
			// the low half negated, one accumulating v_pk_add -- the reference's roundings (convolve_base.c:28-40),
			// spelled out because the vectoriser otherwise spends 6-7 instructions per tap on it
			trx_v2f acc = { 0.0f, 0.0f };
			const int base = i + start - (N - 1);
			for (int k0 = 0; k0 < N; k0 += 8) {
				c32 x[8];
#pragma unroll
				for (int u = 0; u < 8; u++) {
					const int j = base + k0 + u;
					if (PADDED) x[u] = lds_c32(sig + j);
					else x[u] = (j >= 0 && j < sig_len) ? sig[j] : make_float2(0.0f, 0.0f);
				}
#pragma unroll
				for (int u = 0; u < 8; u++) {
					// two taps per 16-byte broadcast read (sequence tables are 16-byte aligned in LDS)
					const float4 h2 = reinterpret_cast<const float4 *>(taps + k0)[u >> 1];
					const c32 h = (u & 1) ? make_float2(h2.z, h2.w) : make_float2(h2.x, h2.y);
					const trx_v2f a = (trx_v2f){ x[u].x, x[u].x } * (trx_v2f){ h.x, h.y };
					const trx_v2f b = (trx_v2f){ x[u].y, x[u].y } * (trx_v2f){ h.y, h.x };
					trx_v2f t;                                   // (a.x - b.x, a.y + b.y): the compiler does not fold a
					asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(t) : "v"(a), "v"(b));   // half negation
					acc = acc + t;
				}
			}
			const c32 y = make_float2(acc.x, acc.y);
			cz[i] = y;
			// fastPeakDetect (:1120-1139): first strict maximum of |corr|^2 (per lane: i ascending)
			const float v = norm2(y);
			if (v > best) { best = v; bidx = i; }
		}
		if (lane < TRX_CZ_PAD)
			cz[len + lane] = make_float2(0.0f, 0.0f);    // right zero pad (len varies per burst)
	}

	DIAG_MARK(3);
	// arg-max across lanes: wave max (DPP), then the lowest index holding it (ballot + scalar ff1)
	const float m = wave_max(best);
	if (!(m > 0.0f))
		return 0;                                    // toa = -1 < 3
	if (pair) {                                      // the lag index grows with the lane: the lowest lane holding the maximum
		const unsigned long long hit = __ballot(best == m);
		bidx = __builtin_amdgcn_readlane(bidx, __ffsll((unsigned long long)hit) - 1);
	} else {
		const unsigned long long hit = __ballot(best == m);
		const unsigned long long hit_lo = __ballot(best == m && bidx == lane);
		bidx = hit_lo ? (__ffsll((unsigned long long)hit_lo) - 1) : (64 + __ffsll((unsigned long long)hit) - 1);
	}
	c32 *czp = cz;
	if (wide) {
		c32 *win = cz - TRX_CZ_PAD;                      // 2 * TRX_CZ_PAD + 1 entries: positions bidx - 12 .. bidx + 12
		if (lane < 2 * TRX_CZ_PAD + 1) {
			const int g = bidx - TRX_CZ_PAD + lane;
			win[lane] = (g >= 0 && g < len) ? corr_at(g) : make_float2(0.0f, 0.0f);
		}
		czp = win - (bidx - TRX_CZ_PAD);
	}
#ifdef TRX_WHATIF_PAIR
	wi.bidx = (bidx < 3) ? 3 : (bidx > len - 3 ? len - 3 : bidx);
#endif
	const int r = detect_tail_h<FAST>(sig, sig_len, czp, hdr, N, thresh, start, len, bidx, sincv, pc, lane, toa_out, amp_out, ci_out,
					  on_toa, wa4, slice DIAG_PASS WI_PASS);
	if (wide) {
		wave_sync();
		if (lane < TRX_CZ_PAD)
			cz[lane - TRX_CZ_PAD] = make_float2(0.0f, 0.0f);     // the window covered cz's left zero pad: restore it
	}
	return r;
}

template <bool PADDED, bool NARROW>
__device__ __forceinline__ int detect_burst(const c32 *sig, int sig_len, c32 *cz, const c32 *taps, const float *hdr,
					     int N, float thresh, int start, int len, const float *sincv, const PeakConst &pc, int lane,
					     float *toa_out, c32 *amp_out, float *ci_out, int slice, int unit_slot DIAG_ARG)
{
	WI_LOCAL;
	return detect_burst_h<PADDED, NARROW, false>(sig, sig_len, cz, taps, hdr, N, thresh, start, len, sincv, pc, lane, toa_out, amp_out,
						     ci_out, NoToaHook(), nullptr, slice, unit_slot DIAG_PASS WI_PASS);
}


// ------------------------------------------------------------------------------------------------
// detectAnyBurst() (sigProcLib.cpp:1926-1957) for one burst: up to 3 detectGeneralBurst() windows, first hit wins
// (TSC; EDGE with fall-through to TSC :1933-1941; RACH / EXT_RACH with TS0..TS2 :1788-1800).
//   decimate(lo, hi): make sig[lo..hi) valid (4 SPS: downsampleBurst restricted to what the correlation and
//                     computeCI read, :1587-1601); a no-op at 1 SPS where sig is the burst itself
//   lseq / lhdr     : LDS sequence table and headers (LSEQ_* layout)
// Returns rc exactly as detectAnyBurst(): CorrType (>0) | 0 | -SignalError.  Wave-uniform.
// ------------------------------------------------------------------------------------------------
struct DetectOut { float toa; c32 amp; float ci; int tsc; };

//   unit_bad        : set by decimate() when a sample it produced fails unit_unsafe() (or -1-initialised to disable
//                     the addition-only correlation altogether: 1-SPS kernel, table mismatch)
template <bool PADDED, bool NARROW, typename DecimateFn>
__device__ __forceinline__ int detect_any_burst(int type, int tsc, int max_toa, int clip, DecimateFn decimate,
						 const c32 *sig, int sig_len, c32 *cz, const c32 *lseq, const float *lhdr,
						 float thresh, const float *sincv, const PeakConst &pkc, int lane, int slice,
						 const int &unit_bad, DetectOut *out DIAG_ARG)
{
	int ncand = 0;
	if (type == TRXHIP_TSC || type == TRXHIP_EDGE) {
		if (tsc > 7)
			return -TRXHIP_SIGERR_UNSUPPORTED;               // :1893, :1912
		ncand = (type == TRXHIP_EDGE) ? 2 : 1;
	} else if (type == TRXHIP_RACH || type == TRXHIP_EXT_RACH) {
		ncand = (type == TRXHIP_EXT_RACH) ? 3 : 1;           // :1791
	} else if (type == TRXHIP_IDLE && (slice & TRXHIP_FLAG_IDLE_DUMMY)) {
		ncand = 1;                                           // detectDummyBurst (:1863-1877, :1945-1947)
	}                                                        // other types: "Invalid correlation type", rc = 0 (:1949-1950)
	if (ncand > 0 && max_toa > TRXHIP_MAX_TOA)
		return -TRXHIP_SIGERR_UNSUPPORTED;                   // this implementation's window limit (trxhip.h)

	int dec_lo = 1 << 30, dec_hi = 0;                        // range of sig[] already valid
	for (int c = 0; c < ncand; c++) {
		// one detectGeneralBurst() call (:1732-1771): sequence + window
		int slot, target, head, tail, N;
		if (type == TRXHIP_RACH || type == TRXHIP_EXT_RACH) {
			slot = 8 + c; target = 48; head = 8; tail = 8 + max_toa; N = 40;       // :1788-1790
		} else if (type == TRXHIP_EDGE && c == 0) {
			slot = 11 + tsc; target = 82; head = 6; tail = 6 + max_toa; N = 16;    // :1915-1918
		} else if (type == TRXHIP_IDLE) {
			slot = 19; target = 82; head = 10; tail = 6 + max_toa; N = 16;         // :1869-1872
		} else {
			slot = tsc; target = 82; head = 10; tail = 6 + max_toa; N = 16;        // :1896-1899
		}
		const c32 *taps = lseq + ((slot < 8) ? LSEQ_TSC(slot) : (slot < 11) ? LSEQ_RACH(slot - 8) : (slot < 19) ? LSEQ_EDGE(slot - 11) : LSEQ_DUMMY);
		const float *hdr = lhdr + 8 * slot;
		const int start = target - head - 1;                 // :1752
		const int len = head + tail;                         // :1753

		int lo = start - (N - 1); if (lo < 0) lo = 0;
		int hi = start + len;     if (hi > sig_len) hi = sig_len;
		if (lo < dec_lo || hi > dec_hi) {
			decimate(lo, hi);
			dec_lo = lo; dec_hi = hi;
		}
		DIAG_MARK(2);
		float t; c32 a; float cc;
		const int unit_slot = (PADDED && !unit_bad && (slot < 11 || slot == 19)) ? slot : -1;
		const int hit = detect_burst<PADDED, NARROW>(sig, sig_len, cz, taps, hdr, N, thresh, start, len, sincv, pkc, lane, &t, &a, &cc,
							     slice, unit_slot DIAG_PASS);
		wave_sync();
		if (hit) {
			out->toa = t - (float)head;                      // :1768
			out->amp = a;
			out->ci = cc;
			if (slot >= 8 && slot < 11) { out->tsc = slot - 8; return type; }    // :1797
			if (slot == 19) { out->tsc = 0; return TRXHIP_IDLE; }                 // :1874, :1953-1954
			out->tsc = tsc;
			return (slot >= 11) ? TRXHIP_EDGE : TRXHIP_TSC;  // :1953-1954
		}
	}
	return (ncand > 0 && clip) ? -TRXHIP_SIGERR_CLIP : 0;    // :1764
}

// ------------------------------------------------------------------------------------------------
// 8-PSK tail of demodEdgeBurst() (sigProcLib.cpp:2105-2128) on the 1-SPS burst dec[0..n_dec) (LDS):
//   eq  = convolve(dec, c0_inv, NO_DELAY)            5 real taps, zero outside            (:2116, :405-422)
//   rot = derotateEdgeBurst(eq, 1)                   x (cosf(p), -sinf(p)), p = (i%16)*3pi/8  (:691-711)
//   ci  = computeEdgeCI(rot)                         EVM against the nearest 8-PSK point   (:2074-2093)
//   soft= softSliceEdgeBurst(rot)                    444 Manhattan soft bits               (:1962-2006)
// Element-wise work in the reference's operand order (bit-exact); the C/I sum is a wave tree sum.
// Writes so[0 .. min(444, soft_stride)) and zero-fills the rest; returns C/I in dB.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_post(const c32 *dec, int n_dec, const trx_tables *__restrict__ tab,
					    float *so, int soft_stride, int slice, int lane)
{
	const c32 r_pi8 = make_float2(tab->edge_rot2[0].re, tab->edge_rot2[0].im);
	const c32 r_pi4 = make_float2(tab->edge_rot2[1].re, tab->edge_rot2[1].im);
	float err = 0.0f;
	for (int i = lane; i < n_dec; i += WAVE) {
		float er = 0.0f, ei = 0.0f;
#pragma unroll
		for (int k = 0; k < 5; k++) {
			const int j = i - 2 + k;
			c32 x = make_float2(0.0f, 0.0f);
			if (j >= 0 && j < n_dec) x = dec[j];
			const float h = tab->c0_inv[k];
			er += x.x * h;
			ei += x.y * h;
		}
		const trx_c32 d = tab->edge_derot[i & 15];
		const c32 rot = cmul(make_float2(er, ei), make_float2(d.re, d.im));
		if (i >= 8 && i < n_dec - 8) {
			// computeEdgeCI's nearest 8-PSK point, k = round(atan2f(y, x) / (pi/4)) (:2081-2084), decided on the octant
			// boundaries |y| = tan(pi/8) |x| instead of through the arc tangent: the same k except for a symbol within the
			// arc tangent's own rounding of a boundary, where both neighbours are equally far and the error sum is the same
			const float ax = fabsf(rot.x), ay = fabsf(rot.y);
			const float b1 = 0.41421356237f * ax, b2 = 0.41421356237f * ay;
			int k = (ay <= b1) ? 0 : (ax <= b2) ? 2 : 1;
			if (rot.x < 0.0f) k = 4 - k;
			if (__builtin_signbit(rot.y)) k = -k;
			// a symbol within 1e-5 of an octant boundary (or on an axis through zero, where the signs of +-0 decide): there the
			// two forms may pick different neighbours -- equally far to 1e-7, but the error sum's last bits would differ.  Those
			// (one symbol in ~1e4) are decided exactly as the reference does
			if (fabsf(ay - b1) <= 1e-5f * ax || fabsf(ax - b2) <= 1e-5f * ay || ax == 0.0f || ay == 0.0f)
				k = (int)roundf(atan2f(rot.y, rot.x) / tab->edge_step);
			const trx_c32 id = tab->edge_ideal[k + 4];
			const c32 e = make_float2(id.re - rot.x, id.im - rot.y);
			err += norm2(e);
		}
		if (i < 148 && so) {
			const c32 a = cmul(rot, r_pi8);                       // rotateBurst2(burst, -M_PI/8)
			float b0 = -a.y, b1 = a.x;
			const c32 q = cmul(make_float2(fabsf(a.x), fabsf(a.y)), r_pi4);   // fold into quadrant 0, rotate by -M_PI/4
			float b2 = -q.y;
			if (slice & 1) {
				b0 = __builtin_amdgcn_fmed3f(0.5f * (b0 + 1.0f), 0.0f, 1.0f);
				b1 = __builtin_amdgcn_fmed3f(0.5f * (b1 + 1.0f), 0.0f, 1.0f);
				b2 = __builtin_amdgcn_fmed3f(0.5f * (b2 + 1.0f), 0.0f, 1.0f);
			}
			if (3 * i + 0 < soft_stride) so[3 * i + 0] = b0;
			if (3 * i + 1 < soft_stride) so[3 * i + 1] = b1;
			if (3 * i + 2 < soft_stride) so[3 * i + 2] = b2;
		}
	}
	if (so)
		for (int i = 444 + lane; i < soft_stride; i += WAVE)
			so[i] = 0.0f;
	err = wave_sum(err);
	return 3.0103f * log2f(1.0f * (float)(n_dec - 16) / err);
}
