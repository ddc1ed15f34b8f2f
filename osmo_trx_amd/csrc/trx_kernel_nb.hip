// trx_kernel_nb.hip -- the NORMAL-BURST kernel: 625-sample int16 bursts at 4 SPS whose slot expects a GMSK normal burst
// (analyzeTrafficBurst, sigProcLib.cpp:1887-1904 -> detectGeneralBurst :1732-1771 -> detectBurst :1649-1709; demodGmskBurst
// :2055-2072 -> demodCommon :2030-2048; vectorSlicer :546-556), fused demodulator + FAST detector, 148 sliced soft bits --
// the call pullRadioVector() makes for a traffic slot and bench.py times.  (Included by trx_kernel4.hip: one translation
// unit, one copy of the device-side counters.)
//
// Same arithmetic as the COMMON instantiation of burst_pull4_kernel (trx_kernel4.hip) -- results are bit-identical to it --
// but nothing else is in here: no access / EDGE / dummy / wide-window / exact-demodulator code, no run-time flags.  A burst
// this kernel cannot finish is FLAGGED (a byte per burst) and left to the general kernel, which is launched behind it over the flags:
//   * the slot is not a normal-burst slot (type != TSC, tsc > 7) or its window is wider than one round (max_toa > 32);
//   * (a decimated sample that fails the addition-only correlation's guard -- about one burst in 3000 -- takes the multiplying form here)
//   * the peak-ratio gate is too close to call for the estimate (about one in 1e5);
//   * the detected TOA is outside the straight-line demodulator's geometry (toa < -0.25 or > 9 symbols).
// What is different from the general kernel, beside what is absent:
//   * soft bits never go through LDS: the three outputs of a lane (symbols 4 + 3 lane + j; symbols 0..3 come from the
//     low-edge lanes) are rotated, scaled and sliced in registers -- the (-j)^i rotation is a quad permutation of ONE
//     per-lane multiplier, applied by the DPP operand of the multiply -- and stored as one 12-byte store per lane (a
//     contiguous 576-byte run per burst), one burst late like the general kernel's (the stores must be older than the
//     prefetch loads: vmcnt retires in order);
//   * a workgroup's static share is one CONTIGUOUS range of bursts (burst = base + item: one scalar add).
#include "trx_k4_common.h"
#include "trx_nb_asm.inc"

#define NB_D_LEN    160                                           /* decimated window (64 used) / parking space of the low-edge tap rows (1 KB) / the
                                                                      156 symbols of the general demodulator (cold) */
#define NB_CZ_LEN   (TRX_CZ_PAD + TRX_CORR_NARROW + TRX_CZ_PAD)
#define NB_SLICE    (K4_XS + NB_D_LEN + NB_CZ_LEN)                /* complex samples per wave */
#define NB_COMP_ROWS (TRX_DELAY_FILTS + 1)
#define NB_TABLES_FLOATS (TRX_SINCV_LDS + 16 * WAVE + NB_COMP_ROWS * 36 + 16 + 8 * 8 + 5 * WAVE + 5 * WAVE + 2 * 8 * 16)
#define NB_TABLES_BYTES (NB_TABLES_FLOATS * 4)
#define NB_WPB 16
#define NB_POOL_RING 64
#define NB_POOL_UNSET (-1)
#define NB_POOL_END (-2)
#define NB_NO_BURST 0xffffffffu
#define NB_LDS_TAIL (16 + 4 * NB_POOL_RING)
#define NB_LDS_BYTES (NB_TABLES_BYTES + NB_WPB * NB_SLICE * 8 + NB_LDS_TAIL)
#define NB_MAX_TOA 32                                             /* window of 16 + max_toa <= 48 lags: 15 + 48 = 63 decimated samples, one per lane */



// Wave priority by phase: a nibble per point of the burst loop (0: loop top, 1: in front of DEC, 2: of DETA, 3: of TAIL, 4: behind
// TAIL), 15 = no instruction.  The demodulator's filter (TAIL) is the one phase that is dense in vector work -- 72 packed FMAs back
// to back; everything else is chains of LDS round trips and scalar glue.  With the filter at priority 0 and the rest at 2, a wave
// that comes out of a wait gets the next issue slot and the filter waves fill what is left: +0.5 % same box (5 rounds, three
// alternatives within +-0.3 % of it: profiles/r06_ab_runs.txt).  Measurement builds: tools/build_variants.py name:-DTRX_NB_PRIO=k.
#ifndef TRX_NB_PRIO
#define TRX_NB_PRIO 0x20fff
#endif
#define NB_PRIO(point) do { if (((TRX_NB_PRIO >> (4 * (point))) & 15) != 15) \
	asm volatile("s_setprio %0" :: "n"((TRX_NB_PRIO >> (4 * (point))) & 3)); } while (0)
typedef float v3f __attribute__((ext_vector_type(3)));
typedef int v4i __attribute__((ext_vector_type(4)));

// byte address of an LDS object (what the ds_* instructions of the asm blocks take)
template <typename T>
__device__ __forceinline__ unsigned lds_addr(const T *p)
{
	return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) T *)p;
}

__global__ void __launch_bounds__(NB_WPB * WAVE)
nb_pull4_kernel(const uint32_t *__restrict__ iq, const trxhip_burst_params *__restrict__ params,
		trxhip_burst_result *__restrict__ results, float *__restrict__ soft, const trx_tables *__restrict__ tab,
		unsigned n_bursts, float thresh, float full_scale, unsigned *__restrict__ pool_ctr, unsigned *__restrict__ redo,
		float gk5, float gk6, float gk7, float gk8)
{
	static_assert(NB_TABLES_BYTES % 16 == 0 && (NB_SLICE * 8) % 16 == 0 && (K4_XS * 8) % 16 == 0 && (NB_D_LEN * 8) % 16 == 0, "16-byte LDS accesses");
	static_assert(NB_LDS_BYTES <= 160 * 1024, "LDS");
	static_assert(NB_ASM_GDEC_OFF == (TRX_SINCV_LDS + 16 * WAVE + NB_COMP_ROWS * 36) * 4, "tools/gen_nb_asm.py: LDS offset of the decimator taps");
	static_assert(NB_ASM_LSEQ_OFF == (NB_TABLES_FLOATS - 2 * 8 * 16) * 4, "tools/gen_nb_asm.py: LDS offset of the training-sequence taps");
	constexpr int NLD = 10;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane0 = threadIdx.x & (WAVE - 1);
	const int wave = uni((int)(threadIdx.x >> 6));

	// ---- LDS carve: [tables][per-wave slices][work counter, pool ring]
	float *const sincv = reinterpret_cast<float *>(smem);           // [4128] swizzled sinc LUT (at LDS offset 0)
	float *const wa4f = sincv + TRX_SINCV_LDS;                     // [4][64] float4: round A's interpolation weights by lane
	const float4 *const wa4 = reinterpret_cast<const float4 *>(wa4f);
	float *const comp = wa4f + 16 * WAVE;                          // [65][36] composite delay-o-decimate filters (shifted by TRX_FUSED_SH)
	float *const gdec = comp + NB_COMP_ROWS * 36;                  // [16] decimator taps
	float *const lhdr = gdec + 16;                                 // [8][8] headers of the eight training sequences
	int *const pkcl = reinterpret_cast<int *>(lhdr + 8 * 8);       // [5][64] PeakConst fields by lane (the exact re-run of the TOA search)
	int *const lcn = pkcl + 5 * WAVE;                              // [5][64] lane constants of the hand-placed blocks (below)
	c32 *const lseq = reinterpret_cast<c32 *>(lcn + 5 * WAVE);     // [8][16] training-sequence taps (the multiplying correlation: guard failures)
	c32 *const wbase = reinterpret_cast<c32 *>(smem + NB_TABLES_BYTES) + (size_t)wave * NB_SLICE;
	int *const wg_next = reinterpret_cast<int *>(reinterpret_cast<c32 *>(smem + NB_TABLES_BYTES) + (size_t)NB_WPB * NB_SLICE);
	int *const pool_g = wg_next + 4;
	c32 *const P = wbase;                                          // polyphase burst: P[r*PH_A + PH_M0 + m] = x[4m + r]
	c32 *const D = wbase + K4_XS;                                  // D[i] = decimated sample 56 + i
	c32 *const cz = D + NB_D_LEN + TRX_CZ_PAD;                     // zero-padded correlation

	// ---- one-time staging of the tables; zero this wave's slice (pads stay zero)
	for (int i = threadIdx.x; i < TRX_SINCV_LDS; i += blockDim.x)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	for (int i = threadIdx.x; i < 16 * WAVE; i += blockDim.x) {
		const int l = i & (WAVE - 1), u = i >> 6;
		const PeakConst pcl = peak_const(l);
		const int q = (u < 8) ? pcl.loA + 512 * (7 - u) : pcl.hiA + 512 * (u - 8);
		wa4f[((u >> 2) * WAVE + l) * 4 + (u & 3)] = (q < TRX_SINCV_LEN) ? tab->sincv[q] : 0.0f;
	}
	for (int i = threadIdx.x; i < NB_COMP_ROWS * 36; i += blockDim.x) {
		const int f = i / 36, j = i % 36;
		comp[i] = (j >= TRX_FUSED_SH) ? tab->comp_filt[f][j - TRX_FUSED_SH] : 0.0f;
	}
	if (threadIdx.x < 16)
		gdec[threadIdx.x] = tab->dec_taps[threadIdx.x];
	if (threadIdx.x < 128)
		lseq[threadIdx.x] = make_float2(tab->seq[TRX_SEQ_TSC0 + (threadIdx.x >> 4)].taps[threadIdx.x & 15].re,
						tab->seq[TRX_SEQ_TSC0 + (threadIdx.x >> 4)].taps[threadIdx.x & 15].im);
	if (threadIdx.x < 64)
		lhdr[threadIdx.x] = reinterpret_cast<const float *>(&tab->seq[TRX_SEQ_TSC0 + (threadIdx.x >> 3)].gain)[threadIdx.x & 7];
	if (threadIdx.x < WAVE) {
		const PeakConst pc0 = peak_const(threadIdx.x);
		pkcl[0 * WAVE + threadIdx.x] = pc0.flA;
		pkcl[1 * WAVE + threadIdx.x] = pc0.loA;
		pkcl[2 * WAVE + threadIdx.x] = pc0.hiA;
		pkcl[3 * WAVE + threadIdx.x] = pc0.offB;
		pkcl[4 * WAVE + threadIdx.x] = pc0.ratio_off;
		// byte offsets relative to &cz[bidx]: the lane's peak-ratio term; round A's first tap (cz + bidx - 1 + flA - 7); round B's offset
		lcn[0 * WAVE + threadIdx.x] = pc0.ratio_off * 8;
		lcn[1 * WAVE + threadIdx.x] = (pc0.flA - 8) * 8;
		lcn[2 * WAVE + threadIdx.x] = pc0.offB;
		// the demodulator's lanes: 0..47 symbols 4 + 3 lane .. with the composite row; 52..55 symbol e = (-lane) & 3, main part of its
		// truncated row (row e of the parked block); 56..59 the same symbols' taps u < 8 (row 4 + e, window two symbols earlier)
		const int l = threadIdx.x, e = (-l) & 3;
		const bool sp = l >= 52 && l < 60;
		lcn[3 * WAVE + l] = 8 * (sp ? (l < 56 ? e : e - 2) : (l < 48 ? 4 + 3 * l : 148));   // first symbol, bytes
		lcn[4 * WAVE + l] = sp ? (e + (l >= 56 ? 4 : 0)) * K4_NTP * 4 : -1;                 // tap row inside the parked block, bytes
	}
	for (int i = lane0; i < NB_SLICE; i += WAVE)
		wbase[i] = make_float2(0.0f, 0.0f);
	if (threadIdx.x == 0) {
		wg_next[0] = NB_WPB;
		wg_next[1] = 0;                                             // the epilogue's election
	}
	if (threadIdx.x < NB_POOL_RING)
		pool_g[threadIdx.x] = NB_POOL_UNSET;
	__syncthreads();

	const float fs_db = 6.02059991f * __log2f(full_scale);          // 20*log10(full_scale)
	// the peak-ratio gate's estimate (tools/gen_nb_asm.py, DETA): thresh^2 / num for the four possible term counts, and the
	// constant term of its certain-pass bound
	const float thr2 = thresh * thresh;
	// (gk5 .. gk8 = thresh^2 / 5 .. thresh^2 / 8 come as kernel arguments: scalar registers)
	const float gc0 = thr2 * 1.0001e-5f;
	// (int)(sync->toa * 512) of the eight training sequences, 4 bits each (+-1 for the reference's tables); a table set whose
	// offsets do not fit leaves every burst to the general kernel
	int t5pk = 0;
	bool t5_ok = true;
	for (int t = 0; t < 8; t++) {
		const int v = (int)(lhdr[8 * t + 5] * 512.0f);
		t5_ok = t5_ok && v >= -8 && v <= 7 && (float)v == lhdr[8 * t + 5] * 512.0f;   // (exact: toa = -nk / 512 then is, too)
		t5pk |= (v & 15) << (4 * t);
	}
	t5pk = uni(t5pk);
	const unsigned long long e8_addr = reinterpret_cast<unsigned long long>(&tab->edge8[0][0][0]);
	// ---- work distribution: groups of 16 consecutive bursts.  Static part: workgroup w owns ONE contiguous range of groups
	// (7/8 of the batch when the cross-die pool is on, everything otherwise); the rest is drawn group by group from a
	// device-wide counter (see burst_pull4_kernel).  Waves claim items one at a time from the workgroup's LDS counter.
	const unsigned n_wg = gridDim.x;
	const unsigned n_groups = (n_bursts + 15u) >> 4;
	const bool pooled = pool_ctr != nullptr;
	const unsigned n_static_groups = pooled ? ((n_groups - (n_groups >> 3)) / n_wg) * n_wg : n_groups;
	const unsigned n_pool_groups = n_groups - n_static_groups;
	const unsigned gq = n_static_groups / n_wg, gr = n_static_groups % n_wg;
	const unsigned my_groups = gq + (blockIdx.x < gr ? 1u : 0u);
	const unsigned base16 = (blockIdx.x * gq + (blockIdx.x < gr ? blockIdx.x : gr)) << 4;   // first burst of this workgroup's range
	unsigned items = my_groups << 4;
	if (base16 + items > n_bursts)
		items = (base16 < n_bursts) ? n_bursts - base16 : 0u;       // the batch's last group may be short
	auto burst_of = [&](unsigned jj) -> unsigned {
		if (jj < items)
			return base16 + jj;
		if (!pooled)
			return NB_NO_BURST;
		const unsigned k = (jj - items) >> 4;
		volatile int *slot = pool_g + (k & (NB_POOL_RING - 1));
		int g;
		while ((g = *slot) == NB_POOL_UNSET)
			__builtin_amdgcn_s_sleep(2);
		g = uni(g);
		if (g < 0)
			return NB_NO_BURST;
		const unsigned bb = ((n_static_groups + (unsigned)g) << 4) + (jj & 15u);
		return bb < n_bursts ? bb : NB_NO_BURST;
	};
	auto pool_draw = [&](unsigned jj, bool ended, int lane) {
		const unsigned k = (jj + 16u - items) >> 4;
		int g = NB_POOL_END;
		if (!ended) {
			unsigned p = 0;
			if (lane == 0)
				p = __hip_atomic_fetch_add(pool_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			p = (unsigned)uni((int)p);
			g = (p < n_pool_groups) ? (int)p : NB_POOL_END;
		}
		if (lane == 0) {
			pool_g[(k + NB_POOL_RING / 2) & (NB_POOL_RING - 1)] = NB_POOL_UNSET;
			pool_g[k & (NB_POOL_RING - 1)] = g;
		}
	};

	// ---- software prefetch of the next burst: ten dwords per lane + the slot's parameters
	uint32_t pre_i[NLD];
	uint32_t pre_prm = 0u;
	auto prefetch = [&](unsigned bb, int lane) {
		pre_prm = reinterpret_cast<const uint32_t *>(params)[2 * (size_t)bb];
		const uint32_t *src = iq + (size_t)bb * 625;
		// (unsigned lane: the offset zero-extends, the loads take the scalar-base form -- no 64-bit vector address whose
		// high half the compiler would build in one of the destination registers, with a wait for vmcnt(0) in front)
		const unsigned ul = (unsigned)lane & 63u;
#pragma unroll
		for (int r = 0; r < NLD - 1; r++)
			pre_i[r] = __builtin_nontemporal_load(src + (unsigned)(r * WAVE) + ul);        // read once: streaming
		// the tenth row has 49 words: lanes 49..63 re-read word 624 (never stored to the LDS).  An exec-masked load behind a
		// "v_mov 0" made the compiler wait for vmcnt(0) -- the nine loads just issued -- in front of the v_mov (measured: -4 %)
		pre_i[NLD - 1] = __builtin_nontemporal_load(src + (unsigned)((NLD - 1) * WAVE) + min(ul, 48u));
	};
	// items j < 16 * my_groups of the static range exist for every workgroup of a launch the launcher sizes (>= 1 group each)
	const unsigned b_first = burst_of((unsigned)wave);
	if (b_first != NB_NO_BURST)
		prefetch(b_first, lane0);

	// ---- deferred output (registers): o = the lane's three sliced soft bits, recw = the result record (lanes 0..7)
	v3f o = { 0.0f, 0.0f, 0.0f };
	int recw = 0;
	bool pend_any = false;
	bool pend_rec = false;                                         // recw holds the pending burst's record (a slot without a burst)
	unsigned pend_b = 0;
	bool left_any = false;
	// ---- the records of detected bursts are made 64 at a time: what computeCI / amp / toa / RSSI need of burst k of the batch
	// goes into lane k of these registers (seven moves under a one-lane mask), flush_records() does the arithmetic once per lane -- the same
	// operations in the same order as block TAIL did per burst on a wave-uniform value (two logarithms and two reciprocals
	// at a quarter of the rate among them), and one 32-byte store per record
	int q_xr = 0, q_xi = 0, q_es = 0, q_toa = 0, q_s = 0, q_fl = 0, q_b = 0;
	int q_n = 0;
	auto flush_records = [&](const int lane) {
		if (lane < q_n) {
			const float *const h = lhdr + 8 * (q_fl & 0xff);        // {gain, 1 / gain, ci_den, toa, n, 1 / ci_den} of the slot's sequence
			const float xr = __int_as_float(q_xr), xi = __int_as_float(q_xi);
			const float a0 = xr * h[2], a1 = xr * h[3], a2 = xi * h[3], a3 = xi * h[2];
			const float amp_re = a0 - a2, amp_im = a1 + a3;         // amp = peak / gain (:1701): peak * (1 / gain), Complex.h:74
			const float toa = (float)q_toa * 0.001953125f;          // position - sync->toa (:1704) - head (:1768): exact in 1/512
			const float energy = __int_as_float(q_es) * 0.0125f;    // energyDetect(burst, 20 * sps): / 80
			const float rssi = fs_db - 3.01029996f * __log2f(energy);
			const float p2 = xi * xi + xr * xr;
			const float C = p2 * h[7];                              // |peak|^2 / ci_den (:1633), table: RN(1 / ci_den)
			const float S = __int_as_float(q_s);
			const float ci = 3.0103f * __log2f(C * __builtin_amdgcn_rcpf(S - C));   // (:1637)
			// (stores the compiler does not see, like every other store of the loop: with its own stores outstanding across the
			// loop's back edge it waits for vmcnt(0) -- the soft bits just stored -- in front of the next prefetch; measured: -3 %)
			const v4i r0 = { 1, __float_as_int(toa), __float_as_int(amp_re), __float_as_int(amp_im) };
			const v4i r1 = { __float_as_int(ci), __float_as_int(energy), __float_as_int(rssi), q_fl };
			asm volatile("global_store_dwordx4 %0, %1, off\n\t"
				     "global_store_dwordx4 %0, %2, off offset:16\n\t"
				     "s_nop 1"
				     :: "v"(results + (unsigned)q_b), "v"(r0), "v"(r1) : "memory");
		}
		q_n = 0;
	};

	// ---- cold: demodGmskBurst for a TOA outside the straight-line geometry (shift w = nk >> 7 above 0: an early burst; below -36:
	// later than 9 symbols).  The general kernel's fused demodulator (trx_kernel4.hip, "FUSED": same sums, same order -- the
	// results are bit-identical to its) on this kernel's buffers: full outputs from the composite filter, the partial ones at the
	// low / high edge from the truncated-composite tables or the masked two-stage sum; the 156 symbols go through D[], from
	// where every lane picks up its three (rotation, slicer: the general kernel's flush()).
	// Returns false for a geometry whose partial outputs the tables do not cover (the general kernel's masked two-stage sum:
	// bursts shorter than the window -- cannot happen with 625 samples and |TOA| within the search windows; such a burst is left to
	// the general kernel).
	auto demod_general = [&](const int nk, const c32 ampv, const int lane) -> bool {
		constexpr int L = 625, nwrite = 148;
		c32 *const dec = D;
		const trx_tables *tabc = tab;                               // (opaque copy: the table addresses of this cold path are formed here,
		asm volatile("" : "+s"(tabc));                              //  not kept in scalar registers across the burst loop)
		const int w = nk >> 7, fr = nk & 127;
		const int fidx = (fr >= 2) ? (fr >> 1) : TRX_DELAY_FILTS;
		const float ian = __builtin_amdgcn_rcpf(norm2(ampv));
		const c32 scale = make_float2(ampv.x * ian, -ampv.y * ian);
		const int n_lo = w > 0 ? w : 0;
		const int n_hi = (L - 1 + w < 623) ? L - 1 + w : 623;
		const int i_full_lo = cdiv4(n_lo + 15), i_full_hi = fdiv4(n_hi);
		const int i0l = cdiv4(n_lo), i0h = i_full_hi + 1;
		const bool need_lo = (n_hi >= n_lo) && (i0l < i_full_lo) && (i0l < nwrite);
		const bool need_hi = (n_hi >= n_lo) && (i0h <= fdiv4(n_hi + 15)) && (i0h < nwrite);
		const bool lo_tab = need_lo && (n_hi >= 4 * (i_full_lo - 1));
		float ct0 = 0.0f, ct1 = 0.0f, ct2 = 0.0f;
		const int le = lane >> 4, lt = lane & 15;
		const int li = i0l + le;
		const int lt0 = n_lo + 15 - 4 * li;
		const bool lact = lt0 >= 1;
		if (lo_tab) {
			const float *row = &tabc->edge_lo[fidx][(lact ? lt0 : 1) - 1][lt];
			ct0 = row[0];
			ct1 = row[16];
			ct2 = (lt < 4) ? row[32] : 0.0f;
		}
		const bool hi_tab = need_hi && (w <= -2) && (n_lo <= 4 * i0h - 15);
		if ((need_lo && !lo_tab) || (need_hi && !hi_tab))
			return false;
		float ch0 = 0.0f, ch1 = 0.0f, ch2 = 0.0f;
		const int hi_i = i0h + le;
		const int htm = n_hi + 15 - 4 * hi_i;
		const bool hact = htm >= 0;
		if (hi_tab) {
			const float *row = &tabc->edge_hi[fidx][hact ? htm : 0][lt];
			ch0 = row[0];
			ch1 = row[16];
			ch2 = (lt < 4) ? row[32] : 0.0f;
		}
		{
			const int c_full = -24 - w;
			const int i_min = cdiv4(-36 - c_full), i_max = fdiv4(L + 1 - c_full);
			int ic = 3 * lane;
			if (ic < i_min) ic = i_min;
			if (ic > i_max - 2) ic = i_max - 2;
			// (the filter as a hand-placed block -- fir24x3's sums in its order -- so that the compiler does not have to find
			// fifty registers for it inside the burst loop)
			v2f acc[3];
			asm volatile(NB_ASM_FIRG
				     : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2])
				     : [nk] "s"(nk), [pb] "s"(lds_addr(P)), [cb] "s"(lds_addr(comp) + 4u * (K4_U0 + TRX_FUSED_SH)), [kic] "v"(8 * ic)
				     : NB_ASM_CLOBBERS);
			if (n_lo == 0 && lo_tab && i_full_hi >= nwrite - 1) {
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
		}
		if (lo_tab) {
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
		wave_sync();
		// this kernel's lanes: 0..47 symbols 4 + 3 lane + j, 52..55 symbol (-lane) & 3; real((-j)^i x) and the slicer as the general
		// kernel's flush() has them (component i & 1, sign by i & 2)
		auto pick = [&](int i) {
			const float d = reinterpret_cast<const float *>(dec + i)[i & 1];
			return __builtin_amdgcn_fmed3f(fmaf((i & 2) ? -0.5f : 0.5f, d, 0.5f), 0.0f, 1.0f);
		};
		const int i0 = (lane < 48) ? 4 + 3 * lane : ((-lane) & 3);
		o.x = pick(i0);
		o.y = pick(lane < 48 ? i0 + 1 : 0);
		o.z = pick(lane < 48 ? i0 + 2 : 0);
		wave_sync();
		return true;
	};

	DIAG_DECL;
	WI_LOCAL;
	unsigned j_next = 0, b_next = NB_NO_BURST;
	for (unsigned b = b_first; b != NB_NO_BURST; b = b_next) {
		int lane;                                                  // re-materialised per burst (see burst_pull4_kernel)
		NB_PRIO(0);
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		const int ticket = claim_issue(wg_next);
		const unsigned prm0 = (unsigned)uni((int)pre_prm);
		// ---- is this a slot the kernel handles?  (a slot of another type is flagged before anything is spent on its samples)
		const unsigned max_toa = prm0 >> 16;
		const int tsc = (prm0 >> 8) & 0xff;
		bool leave = ((prm0 & 0xf8ffu) != (unsigned)TRXHIP_TSC) || (max_toa > NB_MAX_TOA) || !t5_ok;

		// ---- phase 0: registers -> fp32 polyphase LDS; clip scan and energyDetect partial sums on the fly
		c32 *const pload = P + (lane & 3) * PH_A + PH_M0 + (lane >> 2);
		float amax = 0.0f, epart = 0.0f;
#ifdef TRX_DIAG
		DIAG_MARK(12);                                              // (loop top: mbcnt, ticket issue, slot type)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		DIAG_MARK(13);                                              // the prefetched samples' arrival, apart from their conversion
#endif
		if (!leave) {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				if (r < NLD - 1 || i < 625) {
					const c32 v = make_float2((float)(int16_t)(pre_i[r] & 0xffffu), (float)(int16_t)(pre_i[r] >> 16));
					pload[16 * r] = v;
					asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v.x), "v"(v.y));
					if (r < 5)
						epart = fmaf(v.x, v.x, fmaf(v.y, v.y, epart));
				}
			}
		} else {
			// the prefetched samples are dropped, but they must be "used": a path that reaches the next prefetch with loads
			// the compiler still counts as outstanding makes it wait for vmcnt(0) right behind the new loads (measured: -4 %)
#pragma unroll
			for (int r = 0; r < NLD; r++)
				asm volatile("" :: "v"(pre_i[r]));
		}
		DIAG_MARK(14);
		// ---- the previous burst's output: 148 soft bits (lanes 0..47: symbols 4 + 3 lane + j as one 12-byte store, lanes
		// 52..55: symbols 0..3) and the result record (lanes 0..7)
		if (pend_any) {
			float *const so = soft + (size_t)pend_b * 148;
			int *const rp = reinterpret_cast<int *>(results + pend_b);
			const float oe = o.x;
			asm volatile("s_bfm_b64 exec, 48, 0\n\t"
				     "global_store_dwordx3 %0, %1, %2 offset:16\n\t"
				     "s_bfm_b64 exec, 4, 52\n\t"
				     "global_store_dword %3, %4, %2\n\t"
				     "s_mov_b64 exec, %8\n\t"
				     "global_store_dword %5, %6, %7\n\t"
				     "s_mov_b64 exec, -1\n\t"
				     "s_nop 0"
				     :: "v"(lane * 12), "v"(o), "s"(so), "v"(((-lane) & 3) * 4), "v"(oe), "v"(lane * 4), "v"(recw), "s"(rp),
				        "s"(pend_rec ? 0xffull : 0ull)
				     : "memory");
		}
		pend_any = false;
		DIAG_MARK(16);
		// (a converted burst has at least five LDS writes behind the ticket's request -- ten rows, at most two per instruction --
		// and nothing else on lgkmcnt: the ticket is there when five are outstanding, the writes drain under the next block)
		j_next = (unsigned)(leave ? claim_take(ticket) : claim_take_behind<5>(ticket));
		b_next = burst_of(j_next);
		if (pooled && (j_next & 15u) == 0u && j_next + 16u >= items)
			pool_draw(j_next, b_next == NB_NO_BURST && j_next >= items, lane);
		DIAG_MARK(17);
		if (b_next != NB_NO_BURST)
			prefetch(b_next, lane);
		DIAG_MARK(15);

		int clip = 0;
		if (!leave) {
			// maxAmplitude() > 30000 (:1711-1722, :1746)
			clip = __ballot(amax > TRX_CLIP_THRESH) != 0ull;

			// ---- detectGeneralBurst window of a normal burst (:1887-1904: target 82, head 10, tail 6 + max_toa -> start 71,
			// len 16 + max_toa): decimated samples 56 .. 70 + len, one per lane
			const int len = 16 + (int)max_toa;
			__builtin_assume(len >= 16 && len <= 16 + NB_MAX_TOA);
			// (hand-placed blocks: tools/gen_nb_asm.py)  decimator of the window + the addition-only correlation's guard
			const unsigned vd_addr = lds_addr(D) + 8u * (unsigned)lane, vcz_addr = lds_addr(cz) + 8u * (unsigned)lane;
			unsigned long long bad;
			NB_PRIO(1);
			asm volatile(NB_ASM_DEC
				     : [bad] "=s"(bad)
				     : [pd] "v"(lds_addr(P + PH_M0 + 52) + 8u * (unsigned)lane), [vd] "v"(vd_addr), [zero] "v"(0), [nact] "s"(15 + len)
				     : NB_ASM_CLOBBERS);
			DIAG_MARK(2);
			{
				// ---- correlation (lane = lag; the twelve lanes behind the window store the right zero pad), arg-max and the
				// energyDetect sum (:1573-1585); the lane constants of the TOA search are fetched in the reductions' wait states
				float v;
				asm volatile(NB_ASM_CORR
					     : [nrm] "=&v"(v)
					     : [vd] "v"(vd_addr), [vcz] "v"(vcz_addr), [len] "s"(len), [tsc] "s"(tsc), [bad] "s"(bad)
					     : NB_ASM_CLOBBERS);
				DIAG_MARK(3);
				int m_bits, es_bits, bidx;
				int kr, ka, kb, kic, ktp;                                   // lane constants (lcn[])
				asm volatile(NB_ASM_AMAX("ds_read_b32 %[kr], %[l4] offset:%c[lc]", "ds_read_b32 %[ka], %[l4] offset:%c[lc]+256",
							 "ds_read_b32 %[kb], %[l4] offset:%c[lc]+512", "ds_read_b32 %[kic], %[l4] offset:%c[lc]+768",
							 "ds_read_b32 %[ktp], %[l4] offset:%c[lc]+1024", "s_nop 0", "s_nop 0", "s_nop 0")
					     "s_waitcnt lgkmcnt(0)"
					     : [m] "=s"(m_bits), [es] "=s"(es_bits), [bidx] "=s"(bidx), [kr] "=&v"(kr), [ka] "=&v"(ka), [kb] "=&v"(kb),
					       [kic] "=&v"(kic), [ktp] "=&v"(ktp)
					     : [nrm] "v"(v), [ep] "v"(epart), [l4] "v"(4 * lane),
					       [lc] "n"((TRX_SINCV_LDS + 16 * WAVE + NB_COMP_ROWS * 36 + 16 + 64 + 5 * WAVE) * 4)
					     : NB_ASM_CLOBBERS);
				DIAG_MARK(4);
				int hit = 0;
				int toa512 = 0;
				if (m_bits != 0) {                                          // fastPeakDetect: a maximum above zero exists (:1120-1139)
					// edge gate, peak-ratio gate, round A of the TOA bisection and its walk (DETA); round B, walk, peak value (DETB)
					int st, e512;
					float km;
					NB_PRIO(2);
					asm volatile(NB_ASM_DETA
						     : [st] "=&s"(st), [e] "=&s"(e512), [km] "=&v"(km)
						     : [bidx] "s"(bidx), [len] "s"(len), [czb] "s"(lds_addr(cz)), [kr] "v"(kr), [ka] "v"(ka), [l16] "v"(16 * lane),
						       [k5] "s"(gk5), [k6] "s"(gk6), [k7] "s"(gk7), [k8] "s"(gk8), [c0] "v"(gc0)
						     : NB_ASM_CLOBBERS);
					DIAG_MARK(5);
					int xr_bits = 0, xi_bits = 0;
					if (st == 1) {
						asm volatile(NB_ASM_DETB
							     : [st] "=&s"(st), [toa] "=&s"(toa512), [xr] "=&s"(xr_bits), [xi] "=&s"(xi_bits)
							     : [e] "s"(e512), [kb] "v"(kb), [czb] "s"(lds_addr(cz)), [km] "v"(km)
							     : NB_ASM_CLOBBERS);
						DIAG_MARK(6);
					}
					if (st == 3) {
						// an uncertified early / late decision on the path: the search again in the reference's operand order
						if (lane == 0)
							atomicAdd(&g_trx_fast_stats[0], 1ull);
						PeakConst pkc;
						pkc.flA = pkcl[0 * WAVE + lane]; pkc.loA = pkcl[1 * WAVE + lane]; pkc.hiA = pkcl[2 * WAVE + lane];
						pkc.offB = pkcl[3 * WAVE + lane]; pkc.ratio_off = pkcl[4 * WAVE + lane];
						c32 xcorr;
						peak_detect_spec(cz, bidx, sincv, pkc, lane, &toa512, &xcorr, wa4);
						toa512 = uni(toa512);
						xr_bits = uni(__float_as_int(xcorr.x));
						xi_bits = uni(__float_as_int(xcorr.y));
						st = 1;
					}
					if (st == 2) {
						leave = true;                                          // the gate is too close to call for the estimate
						if (lane == 0) atomicAdd(&g_trx_fast_stats[2], 1ull);
					}
					if (st == 1) {
						// computeCI, amp, toa, the result record, 1 / amp; then demodGmskBurst of the usual geometry (TAIL)
						hit = 1;
						int ok, s_bits;
						float d0, d1, d2;
						const int t5 = (t5pk << (28 - 4 * tsc)) >> 28;
						NB_PRIO(3);
						asm volatile(NB_ASM_TAIL
							     : [ok] "=&s"(ok), [ssum] "=&s"(s_bits), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2)
							     : [toa] "s"(toa512), [xr] "s"(xr_bits), [xi] "s"(xi_bits), [t5] "s"(t5), [hdrb] "s"(lds_addr(lhdr) + 32u * (unsigned)tsc),
							       [e8lo] "s"((unsigned)e8_addr), [e8hi] "s"((unsigned)(e8_addr >> 32)), [l16] "v"(16 * lane),
							       [vd] "v"(vd_addr), [pb] "s"(lds_addr(P)), [cb] "s"(lds_addr(comp) + 4u * (K4_U0 + TRX_FUSED_SH)), [db] "s"(lds_addr(D)),
							       [kic] "v"(kic), [ktp] "v"(ktp)
							     : NB_ASM_CLOBBERS);
						DIAG_MARK(10);
						NB_PRIO(4);
						if (!ok) {
							// TOA outside the straight-line geometry (an early burst, or one later than 9 symbols): the general form
							if (lane == 0) atomicAdd(&g_trx_fast_stats[3], 1ull);
							const float *const h = lhdr + 8 * tsc;
							const float xr = __int_as_float(xr_bits), xi = __int_as_float(xi_bits);
							const float a0 = xr * h[2], a1 = xr * h[3], a2 = xi * h[3], a3 = xi * h[2];
							const c32 ampv = make_float2(unif(a0 - a2), unif(a1 + a3));     // amp = peak / gain, as block TAIL has it
							if (!demod_general(t5 + 10 * 512 - toa512, ampv, lane))
								leave = true;
						}
						if (!leave) {
							// the record's inputs: lane q_n of the seven registers (toa in 1/512 symbol: position - sync->toa - head)
							asm("s_lshl_b64 exec, 1, %[n]\n\t"
						    "v_mov_b32_e32 %[qxr], %[xr]\n\t"
						    "v_mov_b32_e32 %[qxi], %[xi]\n\t"
						    "v_mov_b32_e32 %[qes], %[es]\n\t"
						    "v_mov_b32_e32 %[qtoa], %[toa]\n\t"
						    "v_mov_b32_e32 %[qs], %[ss]\n\t"
						    "v_mov_b32_e32 %[qfl], %[fl]\n\t"
						    "v_mov_b32_e32 %[qb], %[b]\n\t"
						    "s_mov_b64 exec, -1"
						    : [qxr] "+v"(q_xr), [qxi] "+v"(q_xi), [qes] "+v"(q_es), [qtoa] "+v"(q_toa), [qs] "+v"(q_s), [qfl] "+v"(q_fl),
						      [qb] "+v"(q_b)
						    : [n] "s"(uni(q_n)), [xr] "s"(xr_bits), [xi] "s"(xi_bits), [es] "s"(es_bits), [toa] "s"(toa512 - t5 - 10 * 512),
						      [ss] "s"(s_bits), [fl] "s"((int)((uint32_t)tsc | ((uint32_t)clip << 8) | (37u << 24))), [b] "s"((int)b));
						q_n++;
						}
						if (ok) {
							o.x = __builtin_amdgcn_fmed3f(fmaf(0.5f, d0, 0.5f), 0.0f, 1.0f);     // vectorSlicer: 0.5 * (x + 1), clamped (:546-556)
							o.y = __builtin_amdgcn_fmed3f(fmaf(0.5f, d1, 0.5f), 0.0f, 1.0f);
							o.z = __builtin_amdgcn_fmed3f(fmaf(0.5f, d2, 0.5f), 0.0f, 1.0f);
						}
					}
				}
				if (!hit && !leave) {
					// nothing found: rc (:1764), energy and RSSI (Transceiver.cpp:741,751), zero soft bits
					const float energy = __int_as_float(es_bits) * 0.0125f;
					const float rssi = fs_db - 3.01029996f * __log2f(energy);
					const uint32_t flags = ((uint32_t)clip << 8) | (1u << 16);
					int word = clip ? -TRXHIP_SIGERR_CLIP : 0;
					word = put_lane<1>(word, 0.0f);
					word = put_lane<2>(word, 0.0f);
					word = put_lane<3>(word, 0.0f);
					word = put_lane<4>(word, 0.0f);
					word = put_lane<5>(word, energy);
					word = put_lane<6>(word, rssi);
					word = write_lane<7>(word, (int)flags);
					recw = word;
					o = (v3f){ 0.0f, 0.0f, 0.0f };
				}
				pend_rec = !hit;
			}
		}
		if (leave) {
			// left to the general kernel: the burst's flag byte, and "something was left" (plain stores: a type-mixed batch leaves a
			// million bursts, and a million atomics on one counter cost ten times the batch)
			// ("something was left" once per wave: a million stores to one address serialise in the L2 like a million atomics)
			if (lane == 0) {
				reinterpret_cast<uint8_t *>(redo + TRX_REDO_HDR)[b] = 1;
				if (!left_any)
					__hip_atomic_store(redo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			left_any = true;
			continue;
		}
		pend_b = b;
		pend_any = true;
		if (q_n == WAVE)
			flush_records(lane);
		DIAG_MARK(11);
	}
	// ---- the last burst's output
	if (pend_any) {
		int lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		float *const so = soft + (size_t)pend_b * 148;
		int *const rp = reinterpret_cast<int *>(results + pend_b);
		const float oe = o.x;
		asm volatile("s_bfm_b64 exec, 48, 0\n\t"
			     "global_store_dwordx3 %0, %1, %2 offset:16\n\t"
			     "s_bfm_b64 exec, 4, 52\n\t"
			     "global_store_dword %3, %4, %2\n\t"
			     "s_mov_b64 exec, %8\n\t"
			     "global_store_dword %5, %6, %7\n\t"
			     "s_mov_b64 exec, -1\n\t"
			     "s_nop 0"
			     :: "v"(lane * 12), "v"(o), "s"(so), "v"(((-lane) & 3) * 4), "v"(oe), "v"(lane * 4), "v"(recw), "s"(rp),
			        "s"(pend_rec ? 0xffull : 0ull)
			     : "memory");
	}
	{
		int lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		flush_records(lane);                                       // the records still in the registers
	}
	DIAG_FLUSH();
	if (pooled) {
		__syncthreads();
		int tid0;                                                   // (re-derived: not a mask kept in scalar registers since the prologue)
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid0));
		// one wave of the workgroup -- whichever reaches the LDS counter first; the wave's index is not kept in a scalar
		// register through the burst loop for this
		if (tid0 == 0 && atomicAdd(wg_next + 1, 1) == 0) {
			const unsigned d = __hip_atomic_fetch_add(pool_ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			if (d == gridDim.x - 1u) {
				__hip_atomic_store(pool_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(pool_ctr + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
}
