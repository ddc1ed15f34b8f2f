// trx_kernel4.hip -- the production kernel for 4-SPS bursts of 624..628 samples (the only burst size the
// transceiver produces at 4 SPS: 625, radioInterface.cpp:254-260).  Same algorithm and boundary as the generic
// burst_pull_kernel (trx_kernels.hip); what differs is how the two resources that bound it are spent:
//
//   * LDS: the burst is kept in a POLYPHASE layout -- four arrays P_r[m] = x[4m + r].  Every FIR of the path
//     advances in steps of 4 samples per lane (the /4 decimator) or a multiple of 4 (12 outputs per lane in
//     the fractional-delay filter), so consecutive lanes read consecutive LDS words of one phase array:
//     bank-conflict-free ds_read_b64, where the linear layout was 2- and 4-way conflicted (42 % of all
//     LDS cycles).  The phase of a tap is wave-uniform, so addresses are lane base + compile-time offset.
//   * VALU: two demodulators.
//       EXACT (TRXHIP_FLAG_EXACT_DEMOD): delayVector -> scaleVector -> downsampleBurst as the reference
//         does it, two FIR stages in the reference's operand order: soft bits bit-identical to generic C.
//       FUSED (default): delay(20 taps) o decimate(16 taps) is ONE 35-tap filter evaluated only at the 148/156
//         symbol instants (5.5 k MACs instead of 15 k), with FMA; a lane owns three adjacent symbols so that their
//         windows share LDS reads, and the symbols go through the (by then free) decimation buffer for a
//         coalesced store.  The reference truncates the intermediate
//         signal (zero outside [0, L) after the delay, zero history in front of the decimator): the <= 8
//         outputs whose decimator window straddles those edges come from truncated-composite tables (or, in unusual
//         geometries, the exact masked two-stage sum), so the result differs from the reference only by rounding and the
//         dropped taps: TRXHIP_FUSED_SOFT_ATOL (include/trxhip.h: 1e-5 absolute on full scale 1; bar 1e-4).
//     Detection: rc, TSC and TOA are identical to the reference's in both modes.  EXACT keeps every sum of the reference (amp and
//     C/I bit-exact too); FUSED runs the FAST detector (round 5, trx_device.h peak_detect_fast): FMA interpolation rounds whose
//     every early / late decision is certified by a proven margin or re-run exactly -- amp / C/I within TRXHIP_FAST_AMP_RTOL /
//     TRXHIP_FAST_CI_ATOL_DB (include/trxhip.h).
//   * occupancy: 16 waves per CU (4 per SIMD) -- per-wave LDS is cut to 7.7 KB (NARROW buffers, trx_device.h) and the
//     kernel kept at 128 VGPRs.  A wave is one serial program per burst and a SIMD runs four: what counts is that all
//     four stay resident to the end of the launch (waves CLAIM bursts, they are not dealt them) and the length of a
//     wave's timeline per burst, to which every issue slot -- vector, scalar, wait -- adds the same.
#include <atomic>
#include "trx_device.h"

#include "trx_k4_common.h"

#define K4_CZ_LEN (TRX_CZ_PAD + TRX_CORR_NARROW + TRX_CZ_PAD)
#define K4_SLICE (K4_XS + TRX_DEC_NARROW + K4_CZ_LEN)
#define K4_DROWS (TRX_DELAY_FILTS + 1)                         // + identity row (no fractional filter)
#define K4_PKC_INTS (5 * WAVE)                                  // lane constants of the TOA bisection (PeakConst), one copy per CU
#define K4_TABLES_FLOATS (TRX_SINCV_LDS + K4_DROWS * TRX_DELAY_HLEN + 2 * 160 + 16 + 2 * LSEQ_TAPS + 8 * LSEQ_NHDR + K4_DROWS * 36 + K4_PKC_INTS)
#define K4_TABLES_BYTES (K4_TABLES_FLOATS * 4)
#define K4_POOL_RING 64                     // ring of dynamic-group ids per workgroup (at most 3 groups are in flight at a time)
#define K4_POOL_UNSET (-1)
#define K4_POOL_END (-2)
#define K4_NO_BURST 0xffffffffu
#define K4_LDS_TAIL (16 + 4 * K4_POOL_RING)  // work counter + pool ring behind the per-wave slices


#ifdef TRX_DIAG
extern "C" int trxhip_diag_read(unsigned long long *out, int reset)
{
	static unsigned long long h[TRX_DIAG_WAVES * 24];
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trx_diag), sizeof(h)) != hipSuccess) return -1;
	for (int k = 0; k < 32; k++) out[k] = 0;
	for (int w = 0; w < TRX_DIAG_WAVES; w++)
		for (int k = 0; k < 24; k++) out[k] += h[w * 24 + k];
	if (reset) {
		for (size_t i = 0; i < sizeof(h) / sizeof(h[0]); i++) h[i] = 0;
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_trx_diag), h, sizeof(h)) != hipSuccess) return -1;
	}
	return 0;
}
// raw per-wave records (24 words each; 20..23 = start / end wall clock, HW_ID, XCC_ID of the last launch)
extern "C" int trxhip_diag_read_waves(unsigned long long *out, int n_waves)
{
	if (n_waves > TRX_DIAG_WAVES) n_waves = TRX_DIAG_WAVES;
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trx_diag), (size_t)n_waves * 24 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

__device__ __forceinline__ int fdiv(int a, int b) { const int q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; }   // b > 0
__device__ __forceinline__ int cdiv(int a, int b) { return -fdiv(-a, b); }
// floor / ceiling of a / 4 for any sign: one or two scalar instructions instead of the generic sequence
__device__ __forceinline__ int fdiv4(int a) { return a >> 2; }
__device__ __forceinline__ int cdiv4(int a) { return (a + 3) >> 2; }


// waves per workgroup (one persistent workgroup per CU): 16 = 4 per SIMD for int16 input, both demodulators (<= 128
// VGPRs; 36.6 KB of tables + 16 x 7.7 KB slices + the work counter = 159.7 KB of the 160 KB LDS, which is why the per-wave
// buffers are the NARROW ones); 12 only for complex64 input with the exact demodulator (20 prefetch registers + its filter
// window: 149 VGPRs; the fused complex64 instantiation -- sigProcLib-signature calls, the multi-ARFCN front end's channel
// streams -- fits 128 since round 3 and runs 16 as well).  Throughput
// is (resident waves) / (a wave's time per burst): 12 -> 16 waves was worth 11 % in round 1, and keeping all 16 busy until
// the end of the launch (work claiming, below) another 13 % in round 2.
#define K4_WPB(CF_, EX_) (((CF_) && (EX_)) ? 12 : 16)


// COMMON = the call pullRadioVector() makes (Transceiver.cpp:665-815) and bench.py times: 625-sample int16 bursts, detection
// + demodulation, vectorSlicer applied, rows of 148 soft bits, no diagnostic flags.  Those launch parameters are then
// compile-time constants (the launcher checks them) and the scalar tests, selects and generic store loops they feed
// disappear from the burst loop; every other call takes the general instantiation of the same source.
// LIST: the launch works through the bursts the normal-burst kernel (trx_kernel_nb.hip) left behind -- slots of other types, wide
// windows, the rare bursts its paths do not cover -- marked in a FLAG BYTE PER BURST (redo + TRX_REDO_HDR words; plain stores, no
// atomics: a type-mixed batch leaves a million of them).  redo[0] != 0: something was left; a launch that finds 0 returns at
// once.  A wave scans the flags 256 at a time (one dword per lane; chunks gw, gw + waves, ...), clears what it read and works
// through the set ones; the next burst is known when the current one is prefetched, as in the claiming form.
#define TRX_REDO_HDR 16                   /* words: [0] anything left, [1] workgroups done (this kernel); flag bytes behind them */
// wave priority: the demodulator's filters at 0, everything else at 2 (measured on the normal-burst kernel: trx_kernel_nb.hip;
// here: profiles/r06_ab_runs.txt section 8).  -DTRX_K4_NO_PRIO: measurement build without it.
#ifdef TRX_K4_NO_PRIO
#define K4_PRIO(p)
#else
#define K4_PRIO(p) asm volatile("s_setprio %0" :: "n"(p))
#endif
template <bool CF32, bool EXACT, bool COMMON, bool LIST = false>
__global__ void __launch_bounds__(K4_WPB(CF32, EXACT) * WAVE)
burst_pull4_kernel(const void *__restrict__ iq_, const trxhip_burst_params *__restrict__ params,
		   trxhip_burst_result *__restrict__ results, float *__restrict__ soft_arg,
		   const trx_tables *__restrict__ tab, const float4 *__restrict__ ebp_arg,
		   unsigned n_bursts_arg, int L_arg, float thresh, float full_scale, int soft_stride_arg, int slice_arg,
		   unsigned *__restrict__ pool_ctr, unsigned *__restrict__ redo)
{
	const unsigned n_bursts = n_bursts_arg;
	if (LIST && uni((int)__hip_atomic_load(redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
		return;
	uint32_t *const rflags = LIST ? reinterpret_cast<uint32_t *>(redo + TRX_REDO_HDR) : nullptr;   // four flag bytes per word
	static_assert(!(COMMON && CF32 && EXACT), "complex64 input: the common instantiation exists for the fused demodulator only (16 waves per CU)");
	static_assert(K4_TABLES_BYTES % 16 == 0 && (K4_SLICE * 8) % 16 == 0 && (K4_XS * 8) % 16 == 0 &&
		      ((TRX_DEC_NARROW + TRX_CZ_PAD) * 8) % 16 == 0, "dec[] and cz[] are read / written 16 bytes at a time");
	const int L = COMMON ? 625 : L_arg;
	const int soft_stride = COMMON ? 148 : soft_stride_arg;
	const int slice = COMMON ? TRXHIP_FLAG_SLICE : slice_arg;
	const float4 *const ebp_in = COMMON ? nullptr : ebp_arg;
	float *const soft = soft_arg;
	constexpr int NLD = 10;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = uni((int)(threadIdx.x >> 6));                  // wave-uniform: burst index and its addresses live in SGPRs
	const int waves_per_block = blockDim.x >> 6;

	// Software prefetch of the next burst (see burst_pull_kernel)
	uint32_t pre_i[NLD];
	c32 pre_c[CF32 ? NLD : 1];
	uint32_t pre_prm = 0u;
	auto prefetch = [&](unsigned bb) {
#ifdef TRX_WHATIF_L2INPUT   /* timing only (tools/): every burst reads one of the first 4096 (cache-resident input): what HBM latency costs */
		bb &= 4095u;
#endif
		pre_prm = reinterpret_cast<const uint32_t *>(params)[2 * (size_t)bb];
		if (CF32) {
			const c32 *src = reinterpret_cast<const c32 *>(iq_) + (size_t)bb * L;
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				pre_c[r] = (r < NLD - 1 || i < L) ? src[i] : make_float2(0.0f, 0.0f);
			}
		} else {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(iq_) + (size_t)bb * L;
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				pre_i[r] = (r < NLD - 1 || i < L) ? src[i] : 0u;
			}
		}
	};
	// LIST: the wave's iterator over the flagged bursts.  Chunk c = bursts 256 c .. 256 c + 255, lane l holding the flags of
	// bursts 256 c + 4 l + q in byte q of its word; l_mask = lanes whose byte l_q is set and not yet handed out.
	const unsigned l_chunks = (n_bursts + 255u) >> 8, l_stride = gridDim.x * 16u;
	unsigned l_next_chunk = blockIdx.x * 16u + (unsigned)wave, l_chunk = 0;
	uint32_t l_word = 0u;
	int l_q = 4;
	unsigned long long l_mask = 0ull;
	unsigned l_done = 0u;                                          // bursts this wave was handed
	auto list_next = [&]() -> unsigned {
		for (;;) {
			if (l_mask != 0ull) {
				const int ln = __ffsll((unsigned long long)l_mask) - 1;
				l_mask &= l_mask - 1ull;
				const unsigned bb = (l_chunk << 8) + 4u * (unsigned)ln + (unsigned)l_q;
				if (bb < n_bursts) {
					l_done++;
					return bb;
				}
				continue;
			}
			if (++l_q < 4) {
				l_mask = __ballot(((l_word >> (8 * l_q)) & 0xffu) != 0u);
				continue;
			}
			if (l_next_chunk >= l_chunks)
				return K4_NO_BURST;
			l_chunk = l_next_chunk;
			l_next_chunk += l_stride;
			uint32_t *const wp = rflags + ((size_t)l_chunk << 6) + lane;       // (the flag area is padded to whole chunks)
			l_word = *wp;
			if (l_word != 0u)
				*wp = 0u;                                                       // consumed: the area is all zero again behind this launch
			l_q = -1;
		}
	};
	unsigned b_first = K4_NO_BURST;
	if (LIST) {
		// the first burst's samples are requested BEFORE the tables are staged (the staging then covers their latency: a launch
		// over a few bursts is all latency); a workgroup none of whose waves found a flag leaves before it stages anything
		// (decided on the first chunk of every wave: exact when the batch has at most one chunk per wave)
		b_first = list_next();
		// (workgroup-wide OR through a word at the end of the dynamic LDS -- the pool ring's first entry, unused in this form; the
		// library's __syncthreads_or() would add static LDS to a kernel that uses all 160 KB)
		volatile int *const any_w = reinterpret_cast<volatile int *>(smem + K4_TABLES_BYTES + (size_t)waves_per_block * K4_SLICE * sizeof(c32)) + 4;
		bool wg_any = true;
		if (l_chunks <= l_stride) {
			if (threadIdx.x == 0)
				*any_w = 0;
			__syncthreads();
			if (b_first != K4_NO_BURST && lane == 0)
				*any_w = 1;
			__syncthreads();
			wg_any = *any_w != 0;
			__syncthreads();
		}
		if (!wg_any) {
			if (threadIdx.x == 0) {
				const unsigned d = __hip_atomic_fetch_add(redo + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
				if (d == gridDim.x - 1u) {
					if (pool_ctr)
						__hip_atomic_store(pool_ctr, __hip_atomic_load(redo + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
								   __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(redo + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(redo, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(redo + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			return;
		}
		if (b_first != K4_NO_BURST)
			prefetch(b_first);
	}

	// ---- LDS carve: [tables][per-wave slices]
	float *sincv = reinterpret_cast<float *>(smem);                 // [4128] swizzled sinc LUT
	float *dfilt = sincv + TRX_SINCV_LDS;                          // EXACT: [65][20] fractional-delay filters + identity;
	                                                               // fused: [4][64] float4, round A's interpolation weights by lane
	const float4 *const wa4 = EXACT ? nullptr : reinterpret_cast<const float4 *>(dfilt);
	c32 *rrot = reinterpret_cast<c32 *>(dfilt + K4_DROWS * TRX_DELAY_HLEN);   // [160] reverse rotation
	float *gdec = reinterpret_cast<float *>(rrot + 160);           // [16] decimator taps
	c32 *lseq = reinterpret_cast<c32 *>(gdec + 16);                // [376] training sequences
	float *lhdr = reinterpret_cast<float *>(lseq + LSEQ_TAPS);     // [20][8] sequence headers
	float *comp = lhdr + 8 * LSEQ_NHDR;                            // [65][36] composite delay-o-decimate filters
	int *pkcl = reinterpret_cast<int *>(comp + K4_DROWS * 36);     // [5][64] PeakConst fields by lane
	c32 *wbase = reinterpret_cast<c32 *>(smem + K4_TABLES_BYTES) + (size_t)wave * K4_SLICE;
	int *const wg_next = reinterpret_cast<int *>(reinterpret_cast<c32 *>(smem + K4_TABLES_BYTES) + (size_t)waves_per_block * K4_SLICE);   // work counter
	int *const pool_g = wg_next + 4;                               // [64] ring: pool group claimed for this workgroup's k-th dynamic group
	c32 *const P = wbase;                                          // polyphase burst: P[r*PH_A + PH_M0 + m] = x[4m + r]
	c32 *const dec = wbase + K4_XS;                                // 1-SPS (decimated) burst, zero tail
	c32 *const cz = dec + TRX_DEC_NARROW + TRX_CZ_PAD;             // zero-padded correlation

	// ---- one-time staging (workgroup-wide) of every table; zero this wave's slice (pads stay zero)
	for (int i = threadIdx.x; i < TRX_SINCV_LDS; i += blockDim.x)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	if (EXACT) {
		for (int i = threadIdx.x; i < K4_DROWS * TRX_DELAY_HLEN; i += blockDim.x)
			dfilt[i] = (i < TRX_DELAY_FILTS * TRX_DELAY_HLEN) ? (&tab->delay_filt[0][0])[i]
									  : ((i - TRX_DELAY_FILTS * TRX_DELAY_HLEN) == 9 ? 1.0f : 0.0f);
	} else {
		// the fused kernel's hot path never needs a delay-filter row (its cold edge rounds read them from global memory):
		// the space holds the 16 sinc weights of every lane's round-A position (peak_detect_spec), tap order fl-7 .. fl+8
		for (int i = threadIdx.x; i < 16 * WAVE; i += blockDim.x) {
			const int l = i & (WAVE - 1), u = i >> 6;
			const PeakConst pcl = peak_const(l);
			const int q = (u < 8) ? pcl.loA + 512 * (7 - u) : pcl.hiA + 512 * (u - 8);
			dfilt[((u >> 2) * WAVE + l) * 4 + (u & 3)] = (q < TRX_SINCV_LEN) ? tab->sincv[q] : 0.0f;
		}
	}
	for (int i = threadIdx.x; i < K4_DROWS * 36; i += blockDim.x) {    // (rows shifted by TRX_FUSED_SH: tap K4_U0 at a multiple of 4)
		const int f = i / 36, j = i % 36;
		comp[i] = (j >= TRX_FUSED_SH) ? tab->comp_filt[f][j - TRX_FUSED_SH] : 0.0f;
	}
	for (int i = threadIdx.x; i < 160; i += blockDim.x)
		rrot[i] = make_float2(tab->rrot1[i].re, tab->rrot1[i].im);
	if (threadIdx.x < 16)
		gdec[threadIdx.x] = tab->dec_taps[threadIdx.x];
	for (int i = threadIdx.x; i < LSEQ_TAPS; i += blockDim.x) {
		int s, k;
		if (i < 128)      { s = TRX_SEQ_TSC0 + i / 16;          k = i % 16; }
		else if (i < 248) { s = TRX_SEQ_RACH0 + (i - 128) / 40; k = (i - 128) % 40; }
		else if (i < 376) { s = TRX_SEQ_EDGE0 + (i - 248) / 16; k = (i - 248) % 16; }
		else              { s = TRX_SEQ_DUMMY;                  k = i - 376; }
		lseq[i] = make_float2(tab->seq[s].taps[k].re, tab->seq[s].taps[k].im);
	}
	for (int i = threadIdx.x; i < 8 * LSEQ_NHDR; i += blockDim.x) {
		const int slot = i / 8;
		const int s = (slot < 8) ? TRX_SEQ_TSC0 + slot : (slot < 11) ? TRX_SEQ_RACH0 + (slot - 8) : (slot < 19) ? TRX_SEQ_EDGE0 + (slot - 11) : TRX_SEQ_DUMMY;
		lhdr[i] = reinterpret_cast<const float *>(&tab->seq[s].gain)[i % 8];
	}
	if (threadIdx.x < WAVE) {
		// functions of the lane id only: kept in LDS and re-read per burst (5 ds_read_b32) rather than in 5 VGPRs that
		// would be live across the whole burst loop (the demodulator's main filter has no registers to spare)
		const PeakConst pc0 = peak_const(threadIdx.x);
		pkcl[0 * WAVE + threadIdx.x] = pc0.flA;
		pkcl[1 * WAVE + threadIdx.x] = pc0.loA;
		pkcl[2 * WAVE + threadIdx.x] = pc0.hiA;
		pkcl[3 * WAVE + threadIdx.x] = pc0.offB;
		pkcl[4 * WAVE + threadIdx.x] = pc0.ratio_off;
	}
	for (int i = lane; i < K4_SLICE; i += WAVE)
		wbase[i] = make_float2(0.0f, 0.0f);
	if (threadIdx.x == 0)
		*wg_next = waves_per_block;                                 // items 0 .. waves-1 are the waves' first ones
	if (threadIdx.x < K4_POOL_RING)
		pool_g[threadIdx.x] = K4_POOL_UNSET;
	__syncthreads();

	const float fs_db = 6.02059991f * __log2f(full_scale);          // 20*log10(full_scale)
	// Work distribution.  The workgroup owns every gridDim.x-th group of 16 bursts ("items" j = 0, 1, ... in that order) and its
	// waves CLAIM them one at a time from an LDS counter, one burst ahead (at prefetch time).  A static split -- the same number
	// of bursts for every wave -- leaves the CU under-occupied for the last third of the kernel: the SIMD's issue arbitration
	// favours its oldest wave, which then finishes its share at 65 % of the kernel's duration, the next one at 75 % ...
	// (tools/wave_timeline.py: wave end times 1200 .. 1860 us inside every workgroup), and four waves per SIMD is exactly what
	// hides this kernel's latencies.
	const unsigned n_wg = gridDim.x;
	// items are handed out in groups of 16 CONSECUTIVE bursts (group g belongs to workgroup g % gridDim.x): neighbouring
	// bursts share the 128-byte line their boundary falls in, and with them on one CU that line is fetched from HBM once
	const unsigned n_groups = (n_bursts + 15u) >> 4;
	// The eight dies do not run this kernel at the same speed (tools/wave_timeline.py, round 3: all workgroups of a die end
	// within 17 us of each other, the dies up to 110 us = 7 % apart, 3.7 % of the chip idle on average).  With a pool counter
	// (large batches) only 7/8 of the groups are dealt statically -- the same number to every workgroup -- and the rest is
	// drawn group by group from ONE device-wide atomic counter by whichever workgroup gets there: the wave that takes the
	// first burst of a group draws the NEXT group (k + 1) and publishes it through the LDS ring pool_g[], so that the global
	// atomic's latency is paid once per 16 bursts by one wave, a burst-time before anybody needs the answer.
	const bool pooled = !LIST && pool_ctr != nullptr;
	const unsigned n_static_groups = pooled ? ((n_groups - (n_groups >> 3)) / n_wg) * n_wg : n_groups;
	const unsigned n_pool_groups = n_groups - n_static_groups;
	const unsigned my_groups = (blockIdx.x < n_static_groups) ? (n_static_groups - blockIdx.x + n_wg - 1) / n_wg : 0u;
	unsigned items = my_groups << 4;                               // statically owned items of this workgroup
	if (!pooled && my_groups && (my_groups - 1) * n_wg + blockIdx.x == n_groups - 1)
		items -= (n_groups << 4) - n_bursts;                        // the batch's last group may be short
	// burst index of item jj, or K4_NO_BURST when the work has run out (dynamic items wait for their group to be published)
	auto burst_of = [&](unsigned jj) -> unsigned {
		if (jj < items)
			return (((jj >> 4) * n_wg + blockIdx.x) << 4) + (jj & 15u);
		if (!pooled)
			return K4_NO_BURST;
		const unsigned k = (jj - items) >> 4;
		volatile int *slot = pool_g + (k & (K4_POOL_RING - 1));
		int g;
		while ((g = *slot) == K4_POOL_UNSET)
			__builtin_amdgcn_s_sleep(2);
		g = uni(g);
		if (g < 0)
			return K4_NO_BURST;
		const unsigned bb = ((n_static_groups + (unsigned)g) << 4) + (jj & 15u);
		return bb < n_bursts ? bb : K4_NO_BURST;
	};
	// the wave holding the first item of (static or dynamic) group kk draws dynamic group kk + 1; `ended`: its own group is
	// already past the end of the pool -- publish that, no atomic.  Also clears the ring entry half a ring ahead.
	auto pool_draw = [&](unsigned jj, bool ended, int lane) {
		const unsigned k = (jj + 16u - items) >> 4;
		int g = K4_POOL_END;
		if (!ended) {
			unsigned p = 0;
			if (lane == 0)
				p = __hip_atomic_fetch_add(pool_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			p = (unsigned)uni((int)p);
			g = (p < n_pool_groups) ? (int)p : K4_POOL_END;
		}
		if (lane == 0) {
			pool_g[(k + K4_POOL_RING / 2) & (K4_POOL_RING - 1)] = K4_POOL_UNSET;
			pool_g[k & (K4_POOL_RING - 1)] = g;
		}
	};

	if (!LIST) {
		b_first = burst_of((unsigned)wave);                         // (static: pooled launches give every workgroup >= 7 groups)
		if (b_first != K4_NO_BURST)
			prefetch(b_first);
	}

	// loader address: sample r*64 + lane -> phase lane&3, m = 16r + lane>>2
	c32 *const pload = P + (lane & 3) * PH_A + PH_M0 + (lane >> 2);

	// Deferred output.  vmcnt counts loads and stores alike and retires in order, so a wave that issues its stores at the end
	// of burst k -- after the prefetch loads of burst k+1 -- cannot consume the last prefetched dword before those stores
	// have been acknowledged as well: ~500 cycles per burst spent waiting for a write nothing depends on.  The common
	// outputs (soft bits of the fused demodulator, which sit in dec[] until the next detection; the result record, whose
	// fields are wave-uniform) are therefore written at the top of the NEXT burst, after its samples have been converted
	// and before its own prefetch is issued: at every wait the prefetch loads are then the youngest memory operations.
	int pend_mode = 0;                                              // 0: record only, 1: soft bits from dec[], 2: zero row
	bool pend_any = false;
	float *pend_so = nullptr;
	int pend_nwrite = 0, pend_rc = 0;
	unsigned pend_b = 0;
	uint32_t pend_flags = 0u;
	float pend_toa = 0.0f, pend_ax = 0.0f, pend_ay = 0.0f, pend_ci = 0.0f, pend_energy = 0.0f, pend_rssi = 0.0f;
	int pend_word = 0;
	auto flush = [&](int lane) {
		if (!pend_any)
			return;
		if (pend_mode == 1) {
			// symbols lane, lane + 64, lane + 128: the first two always exist and are always stored (rows of 128
			// floats or more: 148 / 156 / 444 in practice), only the third needs its range checks; one pointer per lane.
			// real(rot[i] * x[i]) (:2066-2068) with rot[i] = (-j)^i up to the table's 1e-16 phase residue: +re, +im, -re, -im
			// for i & 3 = 0..3, and i & 3 = lane & 3 in all three rounds -- so the lane reads the one component it needs
			// (a 4-byte LDS read) and the sign rides on the slicer's multiplier (fused demodulator only: tolerance 1e-5)
			float *const sp = pend_so + lane;
			const float *const dp = reinterpret_cast<const float *>(dec + lane) + (lane & 1);
			const float hs = (lane & 2) ? -0.5f : 0.5f;
#pragma unroll
			for (int r = 0; r < 3; r++) {
				const int i = lane + r * WAVE;
				const int off = (r < 2 || lane < 32) ? r * WAVE : 0;             // symbol 128 + lane (lanes >= 32: discarded below)
				const float d = dp[2 * off];
				float sv;
				if (slice & 1)                                          // vectorSlicer: 0.5 * (x + 1) = fma(0.5, x, 0.5) (scaling by 2 is exact)
					sv = __builtin_amdgcn_fmed3f(fmaf(hs, d, 0.5f), 0.0f, 1.0f);
				else
					sv = (hs + hs) * d;
				if (r < 2) {
					sp[r * WAVE] = sv;
				} else {
					sv = (i < (COMMON ? 148 : pend_nwrite)) ? sv : 0.0f;   // (COMMON: GMSK rows only, vectorSlicer on)
					if (i < soft_stride)
						sp[2 * WAVE] = sv;
				}
			}
			for (int i = lane + 3 * WAVE; i < soft_stride; i += WAVE) // soft_stride > 192: zero tail
				pend_so[i] = 0.0f;
		} else if (pend_mode == 2) {
			for (int i = lane; i < soft_stride; i += WAVE)
				pend_so[i] = 0.0f;
		}
		// result record: 32 bytes, one dword per lane 0..7.  Every field is wave-uniform: v_writelane drops it into
		// its lane (one instruction per field instead of a compare and a select)
		if (lane < 8)                                               // (assembled when the burst ended: one register waits, not eight)
			reinterpret_cast<int *>(results + pend_b)[lane] = pend_word;
		pend_any = false;
	};

	DIAG_DECL;
#ifdef TRX_WHATIF_PAIR
	WhatIf wi = { 8, 1 };                                          /* (the first normal burst of a wave runs in full) */
	WhatIf wi_full = { 0, 0 };
#endif
	unsigned j_next = 0, b_next = K4_NO_BURST;
	for (unsigned b = b_first; b != K4_NO_BURST; b = b_next) {
		// Re-materialise the lane id per burst (2 VALU ops): otherwise every lane-derived address, tree-node
		// offset and LUT base of every phase is hoisted out of this loop and kept live across it, which
		// costs ~50 VGPRs and spills.
		int lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		const int ticket = LIST ? 0 : claim_issue(wg_next);        // this wave's next item; taken at prefetch time below
		const unsigned prm0 = (unsigned)uni((int)pre_prm);
		DIAG_MARK(13);
		const int type = prm0 & 0xff;
		const int tsc = (prm0 >> 8) & 0xff;
		const int max_toa = prm0 >> 16;
#ifdef TRX_HOT_ONLY   /* reading aid (tools/): the instruction stream of a normal burst alone; not a product build */
		__builtin_assume(type == TRXHIP_TSC);
		__builtin_assume(tsc < 8);
		__builtin_assume(max_toa == 3);
#endif

		int rc = 0;
		float toa = 0.0f, ci = 0.0f, energy = 0.0f, rssi = 0.0f;
		c32 amp = make_float2(0.0f, 0.0f);
		int out_tsc = 0, clip = 0, idle = 1, nbits = 0;
		float *so = (COMMON || soft) ? soft + (size_t)b * soft_stride : nullptr;
		// Fused demodulator, usual geometry (COMMON launches; 0 <= TOA <= 9 symbols, GMSK): the four low-edge outputs are
		// computed INSIDE the main filter loop by eight otherwise idle lanes, each with its own tap row (trx_tables.edge8).
		// The rows (768 bytes for the burst's delay filter) are fetched from L2 as soon as detection knows the TOA -- behind
		// computeCI -- and parked in dec[0..95] once computeCI has read its samples.
		float4 fast_rows = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		int fast_nk = 1 << 30;                                       // -TOA in 1/512 symbol when the fast path applies
		auto fast_fetch = [&](int k512) {
			if (!COMMON || EXACT)
				return;
			const int nk = -k512;
			if ((unsigned)(-(nk >> 7)) > 36u)                        // w = nk >> 7 in [-36, 0]
				return;
			fast_nk = nk;
			const int fr = nk & 127;
			const int fidx = (fr >= 2) ? (fr >> 1) : TRX_DELAY_FILTS;
			if (lane < 2 * K4_NTP)
				fast_rows = reinterpret_cast<const float4 *>(&tab->edge8[fidx][0][0])[lane];
		};
		// The straight-line fused demodulator of the usual geometry (called below, behind detection).
		// (Round 3 also ran access bursts through an extended form of it -- TOA up to 72 symbols, high-side partial outputs
		// from trx_tables.edge_hi, called from the access-burst branch: + 3 % on access bursts over the general path with
		// edge_hi, but the extra live values moved the normal burst's register allocation: + 12 vector / + 10 scalar
		// instructions per normal burst, - 2 %.  Not kept; profiles/r03_ab_runs.txt.)
		auto fast_demod = [&]() {
			// ================= FUSED, usual geometry, straight-line =================
			// (the general form below, with n_lo = 0, four low-edge outputs, no high-edge output among the 148 stored, every
			// window inside the padded arrays: nothing left to decide per burst but the delay filter and the shift)
			const int nk = fast_nk;
			const int w = nk >> 7;                                      // integer shift, -36 .. 0
			const int fr = nk & 127;
			const int fidx = (fr >= 2) ? (fr >> 1) : TRX_DELAY_FILTS;   // delay filter row (64 = none)
			const float ian = __builtin_amdgcn_rcpf(norm2(amp));
			const c32 scale = make_float2(amp.x * ian, -amp.y * ian);   // 1 / amp (Complex.h:75,144-150), 1-ulp reciprocal
			nbits = 148;
			idle = 0;
			// park the low-edge rows: lane l < 48 holds floats 4l .. 4l+3 of the 8 x 24 block
			float *const stage = reinterpret_cast<float *>(dec);
#ifdef TRX_WHATIF_NOFETCHWAIT   /* timing only (tools/): what the wait for the edge8 rows costs -- the rows are not used */
			if (lane < 2 * K4_NTP)
				*reinterpret_cast<float4 *>(stage + 4 * lane) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
			if (lane < 2 * K4_NTP)
				*reinterpret_cast<float4 *>(stage + 4 * lane) = fast_rows;
#endif
			wave_sync();
			// lanes 0..49: outputs 3l .. 3l+2 with the burst's composite row; lanes 52..55: output l - 52, main part of its
			// truncated row; lanes 56..59: the same outputs' taps u < 8 (window 8 samples = 2 outputs earlier); the rest idle
			const bool sp = (lane >= 52) && (lane < 60);
			const float *const tp = sp ? stage + (lane - 52) * K4_NTP : comp + fidx * 36 + (K4_U0 + TRX_FUSED_SH);
			int ic = (lane < 50) ? 3 * lane : 150;
			if (sp) ic = (lane < 56) ? lane - 52 : lane - 58;
			const int c = -24 - w + K4_U0;                              // tap u = K4_U0 of output i reads sample 4i + c
			const PhBase pb = ph_bases(P, c & 3, ic + (c >> 2));
			v2f acc[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
			if (!ABL(5))
				fir24x3(pb, reinterpret_cast<const float4 *>(tp), acc);
			// low-edge outputs: main part (lane 52 + i) + taps u < 8 (lane 56 + i), row_shl:4 inside the last row of 16
			float er = acc[0].x, ei = acc[0].y;
			asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
				     "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf" : "+v"(er), "+v"(ei));
			wave_sync();                                                // (every lane has read its taps from dec[0..95])
			// Only ONE component of every symbol is ever used: flush() stores +-re for even, +-im for odd symbols (the (-j)^i
			// rotation) and reads exactly that float of dec[i].  So the product with 1 / amp is formed for that component alone --
			// a * p + b * q with (p, q) = (s.x, -s.y) for the real, (s.y, s.x) for the imaginary part: a multiply and an fma
			// instead of three packed instructions -- and stored as 4 bytes in the slot flush() reads (a ds_write_b32 costs two
			// thirds of a ds_write_b64).  Symbol 3 lane + j is odd when lane + j is.
			{
				const bool odd = (lane & 1) != 0;
				const float p0 = odd ? scale.y : scale.x, q0 = odd ? scale.x : -scale.y;   // j = 0, 2: parity of the lane
				const float p1 = odd ? scale.x : scale.y, q1 = odd ? -scale.y : scale.x;   // j = 1: the other one
				float *const df = reinterpret_cast<float *>(dec + 3 * lane);
				if (lane < 50) {
					df[0 + (odd ? 1 : 0)] = fmaf(acc[0].y, q0, acc[0].x * p0);
					df[2 + (odd ? 0 : 1)] = fmaf(acc[1].y, q1, acc[1].x * p1);
					df[4 + (odd ? 1 : 0)] = fmaf(acc[2].y, q0, acc[2].x * p0);
				}
				if (lane >= 52 && lane < 56)                            // symbols 0 .. 3, after lanes 0, 1 wrote their (partial) versions
					reinterpret_cast<float *>(dec + (lane - 52))[odd ? 1 : 0] = fmaf(ei, q0, er * p0);
			}
			wave_sync();
			pend_mode = 1;
			pend_so = so;
			pend_nwrite = 148;
		};
		bool fast_done = false;

		// ---- phase 0: registers -> fp32 polyphase LDS; clip scan and energyDetect partial sums on the fly
		float amax = 0.0f, epart = 0.0f;
#pragma unroll
		for (int r = 0; r < NLD; r++) {
			const int i = r * WAVE + lane;
			if (r < NLD - 1 || i < L) {
				c32 v;
				if (CF32) v = pre_c[r];
				else v = make_float2((float)(int16_t)(pre_i[r] & 0xffffu), (float)(int16_t)(pre_i[r] >> 16));
				pload[16 * r] = v;
				// maxAmplitude(): one v_max3_f32 with |.| source modifiers per sample (the compiler's tree of v_max / v_max3
				// is 1.5 instructions per sample)
				asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v.x), "v"(v.y));
				if (r < 5)                                          // samples 4i, i < 80 (:1576-1584); masked to lane % 4 == 0 below
					epart = fmaf(v.x, v.x, fmaf(v.y, v.y, epart));      // (tree-summed anyway: tolerance 3e-6, not an ordered sum)
			}
		}
		DIAG_MARK(14);
		flush(lane);                                               // the previous burst's output (its dec[] is still intact)
		pend_mode = 0;
		DIAG_MARK(16);
		if (LIST) {
			b_next = list_next();
		} else {
			j_next = (unsigned)claim_take(ticket);
			DIAG_MARK(17);
			b_next = burst_of(j_next);
			if (pooled && (j_next & 15u) == 0u && j_next + 16u >= items)
				pool_draw(j_next, b_next == K4_NO_BURST && j_next >= items, lane);
		}
		if (b_next != K4_NO_BURST)
			prefetch(b_next);
		DIAG_MARK(15);

		if (type != TRXHIP_OFF) {                        // Transceiver.cpp:704-707
			// maxAmplitude() > 30000 (:1711-1722, :1746): some lane saw a larger component -- a compare and a ballot, no wave max
			clip = __ballot(amax > TRX_CLIP_THRESH) != 0ull;
			epart = wave_sum_quad0(epart);                          // energyDetect partial sums (lanes = 0 mod 4 hold them)
			energy = epart * 0.0125f;                               // energyDetect(burst, 20*sps): / 80
			if (!ABL(2))
				rssi = fs_db - 3.01029996f * __log2f(energy);       // 20*log10(fs/sqrt(e)), Transceiver.cpp:741,751
			wave_sync();
			DIAG_MARK(1);

			if (ebp_in) {
				const float4 e = ebp_in[b];
				rc = (type == TRXHIP_OFF || type == TRXHIP_IDLE) ? 0 : type;
				toa = unif(e.x);
				amp = make_float2(unif(e.y), unif(e.z));
				out_tsc = tsc;
			} else if ((type != TRXHIP_IDLE || (slice & TRXHIP_FLAG_IDLE_DUMMY)) && !ABL(3)) {   // Transceiver.cpp:754-755
				// ---- detectAnyBurst (:1926-1957); decimation on the polyphase layout:
				// dec[i] = sum_k x[4i-15+k] * g[k];  x[4(i-4) + k'] with k' = k+1 -> phase k'&3, m = i-4 + k'>>2
				PeakConst pkc;
				pkc.flA = pkcl[0 * WAVE + lane]; pkc.loA = pkcl[1 * WAVE + lane]; pkc.hiA = pkcl[2 * WAVE + lane];
				pkc.offB = pkcl[3 * WAVE + lane]; pkc.ratio_off = pkcl[4 * WAVE + lane];
				int unit_bad = (slice & TRX_IFLAG_NO_UNIT) ? 1 : 0;     // guard of the addition-only correlation (corr_unit())
				auto decimate = [&](int lo, int hi) {
					bool bad = false;
					for (int i = lo + lane; i < hi; i += WAVE) {
						const c32 *pd = P + PH_M0 + i - 4;
						float yr = 0.0f, yi = 0.0f;
#pragma unroll
						for (int k = 0; k < 16; k++) {
							const c32 x = lds_c32(pd + ((k + 1) & 3) * PH_A + ((k + 1) >> 2));
							const float g = gdec[k];
							yr += x.x * g;
							yi += x.y * g;
						}
						const c32 y = make_float2(yr, yi);
						dec[i] = y;
						bad |= unit_unsafe(y);
					}
					unit_bad |= (__ballot(bad) != 0ull) ? 1 : 0;
					wave_sync();
				};
				if (type == TRXHIP_TSC && tsc < 8 && max_toa <= 33 && !(slice & TRX_IFLAG_NO_SYM)) {
					// The common slot, straight-line: a normal burst is ONE detectGeneralBurst() window (analyzeTrafficBurst,
					// :1887-1904: target 82, head 10, tail 6 + max_toa -> start 71, len 16 + max_toa <= 49), whose 31 + max_toa
					// <= 64 decimated samples dec[56 ..] are one round with lane = sample.  Same functions as the general
					// dispatch below with the constants folded; every lane decimates (samples past the window are real burst
					// samples, written to dec[] entries nothing reads before the demodulator overwrites them).
					const int len = 16 + max_toa;
					__builtin_assume(len >= 16 && len <= 49);
					{
						// only the 15 + len lanes whose samples the window reads are active: the others would move 16 samples
						// each through the LDS and multiply them for nothing -- same instruction count, less LDS time and
						// less power (the kernel runs at its power limit, DESIGN.md 4.1 "Round 5")
						c32 y = make_float2(0.0f, 0.0f);
						if (lane < 15 + len) {
							y = decimate16_sym<!EXACT>(P + PH_M0 + (56 + lane) - 4, gdec);
							dec[56 + lane] = y;
						}
						unit_bad |= (__ballot(lane < 15 + len && unit_unsafe(y)) != 0ull) ? 1 : 0;
						wave_sync();
					}
					DIAG_MARK(2);
					// as soon as the refined position is known: the burst's TOA in 1/512 symbol (position - sync->toa - head,
					// all multiples of 1/512) and, in the usual geometry, the fetch of its low-edge tap rows (fast_fetch)
					const float *const hdr = lhdr + 8 * tsc;
					auto on_toa = [&](int toa512) { fast_fetch(toa512 - (int)(hdr[5] * 512.0f) - 10 * 512); };
#ifdef TRX_WHATIF_PAIR
					wi.skip ^= 1;
#endif
					const int hit = detect_burst_h<true, true, !EXACT>(dec, 156, cz, lseq + LSEQ_TSC(tsc), hdr, 16, thresh, 71, len, sincv,
										   pkc, lane, &toa, &amp, &ci, on_toa, wa4, slice, unit_bad ? -1 : tsc DIAG_PASS WI_PASS);
					wave_sync();
					rc = hit ? TRXHIP_TSC : (clip ? -TRXHIP_SIGERR_CLIP : 0);                 // :1764, :1953-1954
					toa -= 10.0f;                                                              // :1768
					out_tsc = tsc;
				} else if ((type == TRXHIP_RACH || type == TRXHIP_EXT_RACH) && max_toa <= 64 && !(slice & TRX_IFLAG_NO_SYM)) {
					// Access bursts, straight-line as well: detectRACHBurst (:1782-1803) with TS0 only is one window -- target 48,
					// head 8, tail 8 + max_toa -> start 39, len 16 + max_toa <= 80, 40 taps -- over dec[0 .. 39 + len): two
					// decimation rounds (lane, lane + 64), two correlation rounds inside detect_burst().
					const int len = 16 + max_toa;
					__builtin_assume(len >= 16 && len <= 80);
					bool bad = false;
#pragma unroll
					for (int r = 0; r < 2; r++) {
						const int i = lane + r * WAVE;                          // < 128 <= TRX_DEC_NARROW
						if (i < 39 + len) {                                     // (the lanes behind the window stay idle)
							const c32 y = decimate16_sym<!EXACT>(P + PH_M0 + i - 4, gdec);
							dec[i] = y;
							bad |= unit_unsafe(y);
						}
					}
					unit_bad |= (__ballot(bad) != 0ull) ? 1 : 0;
					wave_sync();
					DIAG_MARK(2);
					int hit = detect_burst_h<true, true, !EXACT>(dec, 156, cz, lseq + LSEQ_RACH(0), lhdr + 8 * 8, 40, thresh, 39, len, sincv,
									     pkc, lane, &toa, &amp, &ci, NoToaHook(), wa4, slice, unit_bad ? -1 : 8 DIAG_PASS
#ifdef TRX_WHATIF_PAIR
									     , wi_full
#endif
									     );
					wave_sync();
					out_tsc = 0;                                                               // ebp->tsc = i (:1797)
					if (!hit && type == TRXHIP_EXT_RACH) {
						// extended access bursts: TS1, then TS2 over the same window, first hit wins (:1791-1800)
						for (int c = 1; c < 3 && !hit; c++) {
							hit = detect_burst_h<true, true, !EXACT>(dec, 156, cz, lseq + LSEQ_RACH(c), lhdr + 8 * (8 + c), 40, thresh, 39, len,
											 sincv, pkc, lane, &toa, &amp, &ci, NoToaHook(), wa4, slice, unit_bad ? -1 : 8 + c DIAG_PASS
#ifdef TRX_WHATIF_PAIR
											 , wi_full
#endif
											 );
							wave_sync();
							out_tsc = c;
						}
					}
					rc = hit ? type : (clip ? -TRXHIP_SIGERR_CLIP : 0);                       // :1764, :1797-1800
					toa -= 8.0f;                                                               // :1768
				} else if (!COMMON && type == TRXHIP_EDGE && tsc < 8 && max_toa <= 33 && !(slice & TRX_IFLAG_NO_SYM)) {
					// EDGE slots, straight-line: detectEdgeBurst (:1906-1924: target 82, head 6, tail 6 + max_toa -> start 75,
					// len 12 + max_toa <= 45, 16 taps of the 8-PSK sequence, multiplying correlation) reads dec[60 .. 87 + max_toa);
					// on a miss detectAnyBurst falls through to the GMSK training sequence (:1933-1941), whose window is the normal
					// burst's dec[56 .. 87 + max_toa).  One decimation round with lane = sample 56 + lane serves both.
					const int len = 16 + max_toa;
					__builtin_assume(len >= 16 && len <= 49);
					{
						const c32 y = decimate16_sym<!EXACT>(P + PH_M0 + (56 + lane) - 4, gdec);
						dec[56 + lane] = y;
						unit_bad |= (__ballot(unit_unsafe(y) && lane < 15 + len) != 0ull) ? 1 : 0;
						wave_sync();
					}
					DIAG_MARK(2);
					int hit = detect_burst_h<true, true, !EXACT>(dec, 156, cz, lseq + LSEQ_EDGE(tsc), lhdr + 8 * (11 + tsc), 16, thresh, 75, len - 4,
									     sincv, pkc, lane, &toa, &amp, &ci, NoToaHook(), wa4, slice, -1 DIAG_PASS
#ifdef TRX_WHATIF_PAIR
									     , wi_full
#endif
									     );
					wave_sync();
					if (hit) {
						rc = TRXHIP_EDGE;                                                          // :1953-1954
						toa -= 6.0f;                                                               // :1768
					} else {
						hit = detect_burst_h<true, true, !EXACT>(dec, 156, cz, lseq + LSEQ_TSC(tsc), lhdr + 8 * tsc, 16, thresh, 71, len, sincv,
										 pkc, lane, &toa, &amp, &ci, NoToaHook(), wa4, slice, unit_bad ? -1 : tsc DIAG_PASS
#ifdef TRX_WHATIF_PAIR
										 , wi_full
#endif
										 );
						wave_sync();
						rc = hit ? TRXHIP_TSC : (clip ? -TRXHIP_SIGERR_CLIP : 0);
						toa -= 10.0f;
					}
					out_tsc = tsc;
				} else {
					DetectOut d;
					rc = detect_any_burst<true, true>(type, tsc, max_toa, clip, decimate, dec, 156, cz, lseq, lhdr, thresh, sincv, pkc,
								    lane, slice, unit_bad, &d DIAG_PASS);
					if (rc > 0) { toa = d.toa; amp = d.amp; ci = d.ci; out_tsc = d.tsc; }
				}
			}
		}

#if defined(TRX_SENS_VALU) || defined(TRX_SENS_PLAIN) || defined(TRX_SENS_LDS) || defined(TRX_SENS_SALU) || defined(TRX_SENS_LDSW)
		// measurement builds only (tools/build_variants.py): N extra instructions of one kind per burst, results unused --
		// the slope of throughput against N is what one instruction of that kind costs (or its removal buys)
		{
			v2f sx = { toa, ci }, sy = sx;
			float sp = toa, sq = ci;
			int ss = (int)b;
#ifdef TRX_SENS_VALU
			asm volatile(".rept %c4\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n .endr" : "+v"(sx), "+v"(sy) : "v"(sx), "v"(sy), "n"(TRX_SENS_VALU / 2));
#endif
#ifdef TRX_SENS_PLAIN
			asm volatile(".rept %c2\n v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n .endr" : "+v"(sp), "+v"(sq) : "n"(TRX_SENS_PLAIN / 2));
#endif
#ifdef TRX_SENS_LDS
			{
				const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)sincv + lane * 8;
				asm volatile(".rept %c3\n ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n .endr\n s_waitcnt lgkmcnt(0)" : "=&v"(sx), "=&v"(sy) : "v"(la), "n"(TRX_SENS_LDS / 2) : "memory");
			}
#endif
#ifdef TRX_SENS_LDSW
			{
				// N extra ds_write_b64 into this wave's correlation pad area (cz[64 + ...] is beyond anything a burst reads back;
				// zeros, so the pads stay zero)
				const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) c32 *)(cz + 64) + (lane & 7) * 8;
				const v2f zz = { 0.0f, 0.0f };
				asm volatile(".rept %c2\n ds_write_b64 %0, %1\n .endr" :: "v"(la), "v"(zz), "n"(TRX_SENS_LDSW) : "memory");
			}
#endif
#ifdef TRX_SENS_SALU
			asm volatile(".rept %c1\n s_add_u32 %0, %0, 3\n .endr" : "+s"(ss) : "n"(TRX_SENS_SALU) : "scc");
#endif
			if (sx.x + sy.y + sp + sq == 1.2345e-30f || ss == 0x7fffffff) energy += 1.0f;      // keep the results alive
		}
#endif
		// The burst's result record (wave-uniform fields dropped into lanes 0..7 of one register; written by flush() at the top
		// of the next burst).  Assembled in front of the demodulator when the straight-line one is about to run -- its ~25
		// scalar and vector instructions then sit between the fetch of the low-edge tap rows (issued when the TOA was known)
		// and the first use of those rows (TRX_EARLY_RECORD; the wait for the rows is 1.8 % of the kernel) -- else behind it.
		auto assemble_record = [&]() {
			const bool det = rc > 0;
			pend_flags = (uint32_t)(det ? out_tsc : 0) | ((uint32_t)clip << 8) | ((uint32_t)idle << 16) | ((uint32_t)(nbits / 4) << 24);
			pend_rc = rc;
			pend_toa = det ? toa : 0.0f;
			pend_ax = det ? amp.x : 0.0f;
			pend_ay = det ? amp.y : 0.0f;
			pend_ci = det ? ci : 0.0f;
			pend_energy = energy;
			pend_rssi = rssi;
			{
				int word = pend_rc;
				word = put_lane<1>(word, pend_toa);                     // results of vector arithmetic: still in vector registers
				word = write_lane<2>(word, __float_as_int(pend_ax));
				word = write_lane<3>(word, __float_as_int(pend_ay));
				word = put_lane<4>(word, pend_ci);
				word = put_lane<5>(word, pend_energy);
				word = put_lane<6>(word, pend_rssi);
				word = write_lane<7>(word, (int)pend_flags);
				pend_word = word;
			}
			pend_b = b;
			pend_any = true;
		};
		bool record_done = false;
		// ---- demodAnyBurst -> demodGmskBurst (:2055-2072) ----
		K4_PRIO(0);                                                 // the filters: dense vector work, lowest priority (see trx_kernel_nb.hip)
		if (COMMON && !EXACT && rc == TRXHIP_TSC && fast_nk != (1 << 30) && !fast_done && !ABL(0)) {
#ifndef TRX_LATE_RECORD
			nbits = 148;                                                // (what fast_demod() sets)
			idle = 0;
			assemble_record();
			record_done = true;
#endif
			fast_demod();
			fast_done = true;
		}
		if (fast_done) {
		} else if (rc > 0 && !ABL(0)) {
			// demodCommon (:2030-2048): delayVector(burst, -toa*sps) -> scaleVector(1/amp) -> downsampleBurst
			// delay = -toa * 4 samples: whole = floor(delay), frac = delay - whole, filter floorf(frac * 64) if frac > 0.01
			// (:1049-1057).  A detected TOA is a multiple of 1/512 symbol (bisection step, table toa, integer head), so
			// delay = -K/128 with K = toa * 512 exactly and the split is integer arithmetic on the scalar unit:
			// whole = -K >> 7, frac = (-K & 127) / 128 > 0.01 <=> (-K & 127) >= 2, index = (-K & 127) >> 1.
			// A caller-supplied TOA (demodulation only) can be anything: that path keeps the float sequence.
			int w, fidx;                                                   // integer shift y[n] = fshift[n - w]; filter row (64 = identity)
			if (ebp_in) {
				const float delay = -toa * 4.0f;
				w = uni((int)floorf(delay));
				const float frac = delay - (float)w;
				const bool use_filt = ((double)fabsf(frac) > 1e-2) && !ABL(4);
				fidx = uni(use_filt ? (int)floorf(frac * (float)TRX_DELAY_FILTS) : TRX_DELAY_FILTS);
			} else {
				const int nk = -uni((int)(toa * 512.0f));
				w = nk >> 7;
				const int fr = nk & 127;
				fidx = (fr >= 2 && !ABL(4)) ? (fr >> 1) : TRX_DELAY_FILTS;
			}
			// (complex) 1.0 / amp = (1,0) * amp.inv()   (Complex.h:75,144-150)
			const float an = norm2(amp);
			c32 ainv;
			if (EXACT) {
				ainv = make_float2(amp.x / an, -amp.y / an);
			} else {                                                       // fused demodulator: 1-ulp reciprocal instead of two divisions
				const float ian = __builtin_amdgcn_rcpf(an);
				ainv = make_float2(amp.x * ian, -amp.y * ian);
			}
			// (complex) 1.0 * amp.inv(): the exact variant multiplies it out as the reference does (Complex.h:74-75);
			// the fused one (tolerance 1e-5) takes ainv as is
			const c32 scale = EXACT ? cmul(make_float2(1.0f, 0.0f), ainv) : ainv;
			const bool is_edge = (rc == TRXHIP_EDGE);                     // 8-PSK: all 156 symbols go through LDS to edge_post()
			nbits = is_edge ? 444 : 148;
			idle = 0;
			const int nwrite = (is_edge || !(slice & 1)) ? 156 : 148;

			// delayed samples n exist for n in [n_lo, n_hi]: n >= 0 (zero history in front of the decimator, :1590),
			// 0 <= n - w < L (delayVector zero-fills what it shifts in, :1071-1090), n <= 623 (:78)
			const int n_lo = w > 0 ? w : 0;
			const int n_hi = (L - 1 + w < 623) ? L - 1 + w : 623;

			float hh[TRX_DELAY_HLEN];                                      // delay filter taps (LDS broadcast reads)
			auto load_hh = [&]() {
				if (EXACT) {
					const float4 *hf4 = reinterpret_cast<const float4 *>(dfilt + fidx * TRX_DELAY_HLEN);
#pragma unroll
					for (int q = 0; q < TRX_DELAY_HLEN / 4; q++) {
						const float4 h4 = hf4[q];
						hh[4 * q + 0] = h4.x; hh[4 * q + 1] = h4.y; hh[4 * q + 2] = h4.z; hh[4 * q + 3] = h4.w;
					}
				} else {                                                    // (cold: the edge rounds of unusual geometries)
#pragma unroll
					for (int k = 0; k < TRX_DELAY_HLEN; k++)
						hh[k] = (fidx < TRX_DELAY_FILTS) ? tab->delay_filt[fidx][k] : (k == 9 ? 1.0f : 0.0f);
				}
			};

			if (EXACT) {
				// ================= EXACT: two FIR stages in the reference's operand order =================
				constexpr int R = 12;                                       // outputs per lane: 12*l .. 12*l+11 (52 lanes)
				const int c0 = -w - 9;                                      // sample of tap 0 of output n: n + c0
				int lc = lane;
				const int lc_min = cdiv(-36 - c0, 12), lc_max = fdiv(L + 6 - c0, 12);
				if (lc < lc_min) lc = lc_min;
				if (lc > lc_max) lc = lc_max;
				const PhBase pb = ph_bases(P, c0 & 3, 3 * lc + (c0 >> 2));
				c32 yv[R];
				{
					// fshift[m] = sum_k X(m - 9 + k) * h[k]  (convolve NO_DELAY, 20 real taps; :1060): tap-outer /
					// output-inner with a sliding window; each output still accumulates k = 0..19 in order.  The taps are
					// broadcast LDS reads, a pair per two taps, and the window runs D samples ahead: the footprint is
					// R accumulators + R + D samples + one tap pair, which keeps this kernel at 16 waves per CU like the
					// fused one (it used to hold all 20 taps and ran at 12).
					// Taps 0, 17, 18, 19 are exactly 0.0f in all 64 filters and in the identity row (the sinc LUT is zero beyond
					// 8 pi, sigProcLib.cpp:990-998; tests/test_capi_cpu.py): fl(x * 0) = +-0 and y + (+-0) == y for every finite
					// x (y starts at +0 and can never become -0), so those four steps of the reference's loop change nothing and
					// are skipped: 16 multiply-adds per output instead of 20, same bits.
					constexpr int D = 1, K0 = 1, K1 = 17;
					// Round 4: written with explicit packed instructions.  From the scalar form (yv.x += x.x * h; yv.y += x.y * h) the
					// compiler built this block out of 385 v_add_f32, 147 v_mul_f32, 146 v_pk_mul_f32, 90 v_mov_b32 and 80 s_nop --
					// 868 vector instructions for 192 complex-by-real multiply-adds, and the exact kernel is vector-issue saturated
					// (profiles/r04_workload_counters.txt).  One v_pk_mul_f32 (tap picked by op_sel) + one v_pk_add_f32 per step:
					// product, then sum, k ascending -- the reference's two roundings per step, 384 instructions.
					const float2 *hp2 = reinterpret_cast<const float2 *>(dfilt + fidx * TRX_DELAY_HLEN);
					v2f xr[R + 19];
					v2f ya[R];
#pragma unroll
					for (int j = 0; j < R; j++)
						ya[j] = (v2f){ 0.0f, 0.0f };
#pragma unroll
					for (int j = K0; j < K0 + R - 1 + D; j++) {
						const c32 t = lds_c32(pb.p[j & 3] + (j >> 2));
						xr[j] = (v2f){ t.x, t.y };
					}
					float2 hq = hp2[K0 >> 1];
#pragma unroll
					for (int k = K0; k < K1; k++) {
						const v2f hpair = { hq.x, hq.y };                       // taps 2 (k >> 1) and 2 (k >> 1) + 1
						if (R - 1 + D + k < R + K1 - 1) {
							const c32 t = lds_c32(pb.p[(R - 1 + D + k) & 3] + ((R - 1 + D + k) >> 2));
							xr[R - 1 + D + k] = (v2f){ t.x, t.y };
						}
						// the R products first, then the R sums: a v_pk_add_f32 straight behind the v_pk_mul_f32 it consumes costs a wait
						// state each time (the compiler's own order reused one temporary pair and paid 156 s_nop in this block)
						v2f pr[R];
#pragma unroll
						for (int j = 0; j < R; j++) {
							if (k & 1)
								asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[j]) : "v"(xr[j + k]), "v"(hpair));
							else
								asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(pr[j]) : "v"(xr[j + k]), "v"(hpair));
						}
#pragma unroll
						for (int j = 0; j < R; j++)
							asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ya[j]) : "v"(pr[j]));
						if ((k & 1) && k + 1 < K1)
							hq = hp2[(k + 1) >> 1];
						__builtin_amdgcn_sched_barrier(0);
					}
#pragma unroll
					for (int j = 0; j < R; j++)
						yv[j] = make_float2(ya[j].x, ya[j].y);
				}
#pragma unroll
				for (int j = 0; j < R; j++) {
					const int m = 12 * lane - w + j;
					const bool ok = (lc == lane) && ((unsigned)m < (unsigned)L);
					const c32 v = ok ? yv[j] : make_float2(0.0f, 0.0f);
					yv[j] = cmul(v, scale);                                 // scaleVector (:1198-1205)
				}
				wave_sync();                 // every lane has read its inputs: safe to overwrite in place
				if (lane < 52) {
#pragma unroll
					for (int j = 0; j < R; j++)                             // y[12l + j] -> phase j&3, m = 3l + j>>2
						P[(j & 3) * PH_A + PH_M0 + 3 * lane + (j >> 2)] = yv[j];
				}
				wave_sync();

				if (so || is_edge) {
					const int nloop = is_edge ? 156 : soft_stride;
					for (int i = lane; i < nloop; i += WAVE) {
						float sv = 0.0f;
						if (i < nwrite && !ABL(5)) {
							const c32 *pd = P + PH_M0 + i - 4;
							float yr = 0.0f, yi = 0.0f;
							if (!(slice & TRX_IFLAG_NO_SYM)) {                  // the detector's decimator: same sums, same order, taps in registers
								const c32 yd = decimate16_sym(pd, gdec);
								yr = yd.x;
								yi = yd.y;
							} else {
#pragma unroll
								for (int k = 0; k < 16; k++) {
									const c32 x = lds_c32(pd + ((k + 1) & 3) * PH_A + ((k + 1) >> 2));
									const float g = gdec[k];
									yr += x.x * g;
									yi += x.y * g;
								}
							}
							if (is_edge) {
								dec[i] = make_float2(yr, yi);
								continue;
							}
							const c32 r = rrot[i];
							sv = r.x * yr - r.y * yi;                       // real(rot * x)  (:2066-2068)
							if (slice & 1)
								sv = __builtin_amdgcn_fmed3f(fmaf(0.5f, sv, 0.5f), 0.0f, 1.0f);
						}
						so[i] = sv;
					}
				}
				wave_sync();
				if (is_edge)
					ci = edge_post(dec, 156, tab, so, soft_stride, slice, lane);
			} else {
				// ================= FUSED: one 35-tap composite filter at the symbol instants =================
				// dec[i] = scale * sum_u comp[u] * X(4i - 24 - w + u) for outputs whose 16 decimator inputs all exist
				const int i_full_lo = cdiv4(n_lo + 15), i_full_hi = fdiv4(n_hi);
				const int i0l = cdiv4(n_lo);                              // low-side partial outputs: [i0l, i_full_lo)
				const int i0h = i_full_hi + 1;                              // high-side partial outputs: [i0h, fdiv(n_hi+15,4)]
				const bool need_lo = (n_hi >= n_lo) && (i0l < i_full_lo) && (i0l < nwrite);
				const bool need_hi = (n_hi >= n_lo) && (i0h <= fdiv4(n_hi + 15)) && (i0h < nwrite);

				// Low-side partial outputs from the truncated-composite table (trx_tables.edge_lo): output i0l + e on the 16
				// lanes of row e, lane t of the row taking taps t, t+16, t+32; the table values are fetched HERE, before the
				// main filter, so that their latency is covered by it.  Requires that no high-side truncation touches
				// these outputs (bursts shorter than the window: generic edge_round() below).
				const bool lo_tab = need_lo && (n_hi >= 4 * (i_full_lo - 1)) && (so || is_edge) && !ABL(6);
				float ct0 = 0.0f, ct1 = 0.0f, ct2 = 0.0f;
				const int le = lane >> 4, lt = lane & 15;
				const int li = i0l + le;                                    // this row's output
				const int lt0 = n_lo + 15 - 4 * li;                         // first decimator tap that sees an existing sample
				const bool lact = lt0 >= 1;                                 // (fewer than 4 partial outputs: surplus rows idle)
				if (lo_tab) {
					// all 35 taps here: a truncated decimator no longer cancels the delay filter's tails (taps u < 8 reach 4e-5)
					const float *row = &tab->edge_lo[fidx][(lact ? lt0 : 1) - 1][lt];
					ct0 = row[0];
					ct1 = row[16];
					ct2 = (lt < 4) ? row[32] : 0.0f;
				}

				// High-side partial outputs the same way (trx_tables.edge_hi): a burst shifted left by w <= -2 ends at delayed
				// sample n_hi = L - 1 + w and output i0h + e sees decimator taps t <= tm = n_hi + 15 - 4 (i0h + e) only.  Access
				// bursts live here (TOA up to 63 symbols); the masked two-stage sum of edge_round() was a third of their time.
				const bool hi_tab = need_hi && (w <= -2) && (n_lo <= 4 * i0h - 15) && (so || is_edge) && !ABL(6);
				float ch0 = 0.0f, ch1 = 0.0f, ch2 = 0.0f;
				const int hi_i = i0h + le;                                  // this row's output
				const int htm = n_hi + 15 - 4 * hi_i;                       // last decimator tap that sees an existing sample
				const bool hact = htm >= 0;
				if (hi_tab) {
					const float *row = &tab->edge_hi[fidx][hact ? htm : 0][lt];
					ch0 = row[0];
					ch1 = row[16];
					ch2 = (lt < 4) ? row[32] : 0.0f;
				}

				if (so || is_edge) {
					// ---- main filter.  Lane l owns the three ADJACENT outputs 3l, 3l+1, 3l+2 (52 lanes): their 35-tap
					// windows overlap in 27 samples, so the lane reads 44 samples from LDS instead of 3 x 36 (sample
					// v = u + 4j serves tap u of output j); stride-3 lanes stay bank-conflict-free.  Taps outer, a ring
					// of 16 samples loaded D ahead of use; one sched_barrier per tap keeps order and register footprint.
					// Of the 35 composite taps only u = 9 .. 27 matter: the decimator's passband sees the fractional-delay
					// filter as a pure delay, so comp_f is the decimator shifted by 9 + frac and the taps outside carry
					// < 1.1e-6 (u < 8) of the filter's absolute sum in every one of the 65 rows (tests/test_capi_cpu.py).
					// The filter runs over u = K4_U0 .. K4_U0 + K4_NT - 1 = 6 .. 29 (trx_tables.h): 24 taps instead of 36.
					const float4 *c4 = reinterpret_cast<const float4 *>(comp + fidx * 36 + (K4_U0 + TRX_FUSED_SH));   // broadcast reads
					const int c_full = -24 - w;                                 // sample of tap 0 of output i: 4i + c_full
					const int c = c_full + K4_U0;
					const int i_min = cdiv4(-36 - c_full), i_max = fdiv4(L + 1 - c_full);
					// lanes holding a full output never need the clamp (i_min + 6 <= i_full_lo, i_full_hi + 6 <= i_max):
					// it only keeps the reads of lanes whose outputs are discarded inside the padded arrays
					int ic = 3 * lane;
					if (ic < i_min) ic = i_min;
					if (ic > i_max - 2) ic = i_max - 2;
					const PhBase pb = ph_bases(P, c & 3, ic + (c >> 2));
					v2f acc[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
					if (!ABL(5) && lane < 52)                                   // 52 lanes x 3 outputs = 156; the rest would compute discarded values
						fir24x3(pb, c4, acc);
					DIAG_MARK(10);
					// the 1-SPS symbols go through dec[] (free once detection is done; the 8-PSK tail wants them there
					// anyway): FIR lanes write theirs, the edge rounds overwrite the few partial ones, then every lane
					// reads back lane + 64 r for a coalesced store.  dec[156..159] stay zero for the next detection.
					if (n_lo == 0 && lo_tab && i_full_hi >= nwrite - 1) {
						// the usual geometry (0 <= toa < 9 symbols): outputs 0..3 are the partial ones and lo_tab overwrites them,
						// everything else up to the last symbol stored is a full output -- no per-output selection
						if (lane < 52) {
#pragma unroll
							for (int j = 0; j < 3; j++)
								dec[3 * lane + j] = cmul(make_float2(acc[j].x, acc[j].y), scale);
						}
					} else {
#pragma unroll
						for (int j = 0; j < 3; j++) {
							const int i = 3 * lane + j;
							const bool full = (i >= i_full_lo) && (i <= i_full_hi);
							const c32 d = full ? cmul(make_float2(acc[j].x, acc[j].y), scale) : make_float2(0.0f, 0.0f);
							if (i < 156)
								dec[i] = d;
						}
					}

					// exact masked two-stage sum for 4 consecutive outputs i0..i0+3: lane = 16*e + t computes
					// g[t] * fshift[n - w] for n = 4(i0+e) - 15 + t if that sample exists, one DPP row sums over t
					auto edge_round = [&](int i0) {
						const int e = lane >> 4, t = lane & 15;
						const int n = 4 * (i0 + e) - 15 + t;
						const bool ok = (n >= n_lo) && (n <= n_hi);
						const int s0 = n - w - 9;
						int m0 = s0 >> 2;
						if (m0 < -9) m0 = -9;
						if (m0 > 158) m0 = 158;
						const PhBase pe = ph_bases(P, s0 & 3, m0);
						v2f acc0 = { 0.0f, 0.0f }, acc1 = { 0.0f, 0.0f };     // two chains: even / odd taps
#pragma unroll
						for (int k0 = 0; k0 < TRX_DELAY_HLEN; k0 += 10) {
							c32 x[10];
#pragma unroll
							for (int q = 0; q < 10; q++)
								x[q] = lds_c32(pe.p[(k0 + q) & 3] + ((k0 + q) >> 2));
#pragma unroll
							for (int q = 0; q < 10; q += 2) {
								// taps k0+q (even) and k0+q+1 (odd) share a register pair
								const v2f hp = { hh[k0 + q], hh[k0 + q + 1] };
								acc0 = pk_fma_tap<0>((v2f){ x[q].x, x[q].y }, hp, acc0);
								acc1 = pk_fma_tap<1>((v2f){ x[q + 1].x, x[q + 1].y }, hp, acc1);
							}
							__builtin_amdgcn_sched_barrier(0);
						}
						const v2f accs = acc0 + acc1;
						const float g = ok ? gdec[t] : 0.0f;
						const float sr = row_sum(accs.x * g), si = row_sum(accs.y * g);
						if (t == 0 && (unsigned)(i0 + e) < 156u)
							dec[i0 + e] = cmul(make_float2(sr, si), scale);
					};
					DIAG_MARK(8);
					if (lo_tab) {
						// X(4 li - 24 - w + lt + 16 j), j = 0..2: one phase array, 4 entries apart
						const int s0 = 4 * li - 24 - w + lt;
						const c32 *pp = P + ((s0 & 3) * PH_A + PH_M0 + (s0 >> 2));
						const c32 x0 = lds_c32(pp), x1 = lds_c32(pp + 4), x2 = lds_c32(pp + 8);
						float ar = x0.x * ct0, ai = x0.y * ct0;
						ar = fmaf(x1.x, ct1, ar); ai = fmaf(x1.y, ct1, ai);
						ar = fmaf(x2.x, ct2, ar); ai = fmaf(x2.y, ct2, ai);
						const float sr = row_sum(ar), si = row_sum(ai);
						if (lt == 0 && lact)
							dec[li] = cmul(make_float2(sr, si), scale);
					}
					if (hi_tab) {
						const int s0 = 4 * hi_i - 24 - w + lt;
						const c32 *pp = P + ((s0 & 3) * PH_A + PH_M0 + (s0 >> 2));
						const c32 x0 = lds_c32(pp), x1 = lds_c32(pp + 4), x2 = lds_c32(pp + 8);
						float ar = x0.x * ch0, ai = x0.y * ch0;
						ar = fmaf(x1.x, ch1, ar); ai = fmaf(x1.y, ch1, ai);
						ar = fmaf(x2.x, ch2, ar); ai = fmaf(x2.y, ch2, ai);
						const float sr = row_sum(ar), si = row_sum(ai);
						if (lt == 0 && hact && hi_i < 156)
							dec[hi_i] = cmul(make_float2(sr, si), scale);
					}
					if ((need_lo && !lo_tab) || (need_hi && !hi_tab))
						load_hh();                                          // only now: 20 registers the main filter does not carry
					if (need_lo && !lo_tab && !ABL(6)) edge_round(i0l);
					if (need_hi && !hi_tab && !ABL(6)) edge_round(i0h);
					DIAG_MARK(9);
					wave_sync();

					if (!is_edge && soft_stride >= 2 * WAVE) {
						pend_mode = 1;                                              // written by flush() at the top of the next burst
						pend_so = so;
						pend_nwrite = nwrite;
					} else if (!is_edge) {                                          // short rows: truncated to soft_stride
#pragma unroll
						for (int r = 0; r < 3; r++) {
							const int i = lane + r * WAVE;
							const int ii = i < 159 ? i : 159;
							const c32 d = dec[ii];
							const c32 rr = rrot[ii];
							float sv = rr.x * d.x - rr.y * d.y;
							if (slice & 1)
								sv = __builtin_amdgcn_fmed3f(fmaf(0.5f, sv, 0.5f), 0.0f, 1.0f);
							sv = (i < nwrite) ? sv : 0.0f;
							if (i < soft_stride)
								so[i] = sv;
						}
					}
				}
				if (is_edge) {
					wave_sync();
					ci = edge_post(dec, 156, tab, so, soft_stride, slice, lane);
				}
				wave_sync();
			}
		} else if (so) {
			pend_mode = 2;
			pend_so = so;
		}

		DIAG_MARK(11);
		K4_PRIO(2);
		if (!record_done)
			assemble_record();
		DIAG_MARK(12);
	}
	flush(lane);
	// re-arm the pool's counter pair for the next launch that is handed it: every wave of this workgroup is past its last draw
	// here, and the workgroup that arrives last at the second word zeroes both (no memset in front of a launch: trx_ctx.h)
	if (pooled) {
		__syncthreads();
		if (threadIdx.x == 0) {
			const unsigned d = __hip_atomic_fetch_add(pool_ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			if (d == gridDim.x - 1u) {
				__hip_atomic_store(pool_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(pool_ctr + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	// LIST: the last workgroup to finish re-arms the header for the next launch that is handed it and reports the number of bursts
	// this launch worked through to the host (pool_ctr: in this form a word of pinned host memory; trx_ctx.h, split_backoff)
	if (LIST) {
		if (lane == 0 && l_done)
			__hip_atomic_fetch_add(redo + 2, l_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		if (threadIdx.x == 0) {
			// (every workgroup of the grid is counted: the ones that found nothing counted themselves when they left)
			const unsigned d = __hip_atomic_fetch_add(redo + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			if (d == gridDim.x - 1u) {
				if (pool_ctr)
					__hip_atomic_store(pool_ctr, __hip_atomic_load(redo + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
							   __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(redo + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(redo, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(redo + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	DIAG_FLUSH();
}

// counters of the FAST detector on the current device (trx_device.h: g_trx_fast_stats); synchronises the device
extern "C" int trx_fast_stats_read(unsigned long long *out4, int reset)
{
	if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_trx_fast_stats), 4 * sizeof(unsigned long long)) != hipSuccess)
		return -1;
	if (reset) {
		const unsigned long long z[4] = { 0, 0, 0, 0 };
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_trx_fast_stats), z, sizeof(z)) != hipSuccess)
			return -1;
	}
	return 0;
}

// the sign patterns compiled into corr_unit() against what the table generator derived from the taps (host side)
extern "C" int trx_unit_masks_match(const trx_tables *t)
{
	static const unsigned long long want[12] = { TRX_UNIT_NEG_TSC0, TRX_UNIT_NEG_TSC1, TRX_UNIT_NEG_TSC2, TRX_UNIT_NEG_TSC3,
						     TRX_UNIT_NEG_TSC4, TRX_UNIT_NEG_TSC5, TRX_UNIT_NEG_TSC6, TRX_UNIT_NEG_TSC7,
						     TRX_UNIT_NEG_RACH0, TRX_UNIT_NEG_RACH1, TRX_UNIT_NEG_RACH2, TRX_UNIT_NEG_DUMMY };
	for (int s = 0; s < 12; s++)                                    // TRX_SEQ_TSC0.. = 0..7, RACH 8..10, dummy 11
		if (!((t->unit_ok >> s) & 1u) || t->unit_neg[s] != want[s])
			return 0;
	return 1;
}

#include "trx_kernel_nb.hip"

// The normal-burst kernel over the whole batch, then the general kernel (fused demodulator, common launch parameters) over the
// bursts the first one left behind (d_redo: TRX_REDO_HDR words of header + one flag byte per burst, padded to a multiple of
// 256: all zero on entry and all zero again when the second kernel has finished).  Preconditions (the caller checks them): int16 input of 625 samples,
// fused demodulator, sliced rows of 148 soft bits, tables with the unit / symmetric / FAST structure.
extern "C" int trx_launch_pull4_nb(unsigned *d_pool_ctr, const void *d_iq, const trxhip_burst_params *d_params,
				   trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, size_t n_bursts,
				   float thresh, float full_scale, int n_cu, unsigned *d_redo, unsigned *h_left, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	const size_t need = (n_bursts + 15) / 16;
	size_t grid = (size_t)n_cu;
	if (grid > need) grid = need;
	unsigned *const pool = (d_pool_ctr && grid == (size_t)n_cu && need >= 8 * grid) ? d_pool_ctr : nullptr;
	{
		auto k = nb_pull4_kernel;
		TRX_ARM_DYNAMIC_LDS(k);
		hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NB_WPB * WAVE), NB_LDS_BYTES, stream, reinterpret_cast<const uint32_t *>(d_iq),
				   d_params, d_results, d_soft, d_tab, (unsigned)n_bursts, thresh, full_scale, pool, d_redo,
				   thresh * thresh * 0.2f, thresh * thresh * (1.0f / 6.0f), thresh * thresh * (1.0f / 7.0f), thresh * thresh * 0.125f);
	}
	{
		auto k = burst_pull4_kernel<false, false, true, true>;
		TRX_ARM_DYNAMIC_LDS(k);
		const size_t lds = K4_TABLES_BYTES + (size_t)K4_WPB(false, false) * K4_SLICE * sizeof(c32) + K4_LDS_TAIL;
		hipLaunchKernelGGL(k, dim3((unsigned)n_cu), dim3(K4_WPB(false, false) * WAVE), lds, stream, d_iq, d_params, d_results, d_soft, d_tab,
				   (const float4 *)nullptr, (unsigned)n_bursts, 625, thresh, full_scale, 148, TRXHIP_FLAG_SLICE, h_left, d_redo);
	}
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}

extern "C" int trx_launch_pull4(unsigned *d_pool_ctr, const void *d_iq, int cf32, const trxhip_burst_params *d_params,
				trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, const float *d_ebp_in,
				size_t n_bursts, int L, float thresh, float full_scale, int soft_stride, int flags, int n_cu,
				hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	const bool exact = (flags & TRXHIP_FLAG_EXACT_DEMOD) != 0;      // two kernels: the demodulator is a compile-time choice
	int wpb = K4_WPB(cf32 != 0, exact);
#ifdef TRX_DIAG
	if (const char *e = getenv("TRXHIP_WPB")) { const int v = atoi(e); if (v >= 1 && v <= wpb) wpb = v; }   // occupancy scan
#endif
	const size_t lds = K4_TABLES_BYTES + (size_t)wpb * K4_SLICE * sizeof(c32) + K4_LDS_TAIL;   // + work counter + pool ring
	size_t need = (n_bursts + 15) / 16;                             // work is handed out in groups of 16 bursts
	size_t grid = (size_t)n_cu;
#ifdef TRX_DIAG
	if (const char *e = getenv("TRXHIP_GRID")) { const int v = atoi(e); if (v >= 1) grid = (size_t)v; }
#endif
	if (grid > need) grid = need;
	/* the instantiation with the common launch parameters folded (see the kernel) */
	const bool common = (!cf32 || !exact) && L == 625 && d_soft && !d_ebp_in && soft_stride == 148 &&
			    (flags & ~TRXHIP_FLAG_EXACT_DEMOD) == TRXHIP_FLAG_SLICE;
#define LAUNCH4(CF_, EX_, CM_)                                                                                  \
	do {                                                                                                    \
		auto k = burst_pull4_kernel<CF_, EX_, CM_>;                                                     \
		/* the > 64 KB dynamic-LDS opt-in is per kernel and device: once, not per launch (small batches) */ \
		static std::atomic<unsigned long long> armed{0ull};                                             \
		int dev = 0;                                                                                    \
		if (hipGetDevice(&dev) != hipSuccess) return TRXHIP_EIO;                                        \
		const unsigned long long bit = 1ull << (dev & 63);                                              \
		if (!(armed.load(std::memory_order_acquire) & bit)) {                                           \
			if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,        \
						(int)(K4_TABLES_BYTES + (size_t)K4_WPB(CF_, EX_) * K4_SLICE * sizeof(c32) + K4_LDS_TAIL)) != hipSuccess) \
				return TRXHIP_EIO;                                                                  \
			armed.fetch_or(bit, std::memory_order_release);                                             \
		}                                                                                               \
		hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(wpb * WAVE), lds, stream, d_iq, d_params, d_results, \
				   d_soft, d_tab, reinterpret_cast<const float4 *>(d_ebp_in), (unsigned)n_bursts, L, thresh, \
				   full_scale, soft_stride, flags, pool, (unsigned *)nullptr);                  \
	} while (0)
	/* the cross-die pool needs every workgroup to own >= 7 static groups and the grid to be the persistent one */
	unsigned *const pool = (d_pool_ctr && grid == (size_t)n_cu && need >= 8 * grid) ? d_pool_ctr : nullptr;   /* (TRXHIP_NO_POOL: trx_capi.cpp) */
	if (cf32)        { if (exact) LAUNCH4(true, true, false); else if (common) LAUNCH4(true, false, true); else LAUNCH4(true, false, false); }
	else if (common) { if (exact) LAUNCH4(false, true, true); else LAUNCH4(false, false, true); }
	else             { if (exact) LAUNCH4(false, true, false); else LAUNCH4(false, false, false); }
#undef LAUNCH4
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
