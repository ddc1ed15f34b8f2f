// trx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for osmo-trx's receive-side burst DSP.
//
// Hot kernel: burst_pull_kernel = the DSP core of Transceiver::pullRadioVector()
// (Transceiver52M/Transceiver.cpp:724-803): convert_short_float -> energyDetect -> clip check ->
// detectAnyBurst -> demodAnyBurst -> vectorSlicer, for a batch of independent bursts.
//
// Mapping (MI355X-first, not a translation of the SSE code):
//   * ONE WAVEFRONT (64 lanes) PER BURST, persistent waves grid-striding over the batch, 12 waves per
//     workgroup = one workgroup per CU.  A burst (625 x int16 IQ = 2500 B) is read from HBM exactly once
//     with coalesced dword loads that are software-prefetched one burst ahead, converted to fp32 in
//     flight and kept in that wave's private LDS slice (~8 KB) until its soft bits and 32-byte result
//     record are written: zero intermediate HBM traffic (measured: 1.02 x algorithmic bytes).
//   * every table the path touches (sinc LUT, 64 fractional-delay filters, decimator taps, training
//     sequences, reverse rotation) is staged ONCE per workgroup into LDS (26 KB); wave-uniform taps are
//     LDS broadcast reads, per-lane gathers (sinc LUT) use a bank-swizzled layout.
//   * waves never synchronise with each other after that staging; intra-wave ordering relies on
//     wave-lockstep LDS execution (wave_sync()).
//   * the kernel is VALU-issue bound (not HBM bound: ~26 FLOP/B), so the code is organised to spend
//     vector instructions on the FIR arithmetic only: reductions are DPP (v_max/v_add with row/bcast
//     controls), arg-max and the TOA bisection walk are v_cmp ballots consumed by the scalar unit,
//     range checks are replaced by zero-padded LDS buffers.
//   * peak/TOA: the reference's 9-step early/late bisection (19 data-dependent sinc interpolations) is
//     evaluated SPECULATIVELY: the binary decision tree is expanded across lanes (2 rounds: levels 0-4,
//     then 5-8 + the 16 possible final positions), each lane doing one sequential 16-tap interpolation.
//   * no MFMA: these are short 1-D real/complex convolutions (<= 40 taps).
//
// Numerics: every sum that feeds a DECISION (correlation, peak ratio, bisection compares, filter
// choice) is accumulated in the reference's generic-C order (arch/common/convolve_base.c:28-54) and
// the file is compiled with -ffp-contract=off, so rc / TOA / amp / soft bits are bit-identical to the
// generic-C reference.  Only energyDetect (tree-summed), log2f (C/I) and log10f (RSSI) differ at the
// 1e-6 level.
#include "trx_device.h"

// ------------------------------------------------------------------------------------------------
// the hot kernel
// ------------------------------------------------------------------------------------------------
// (1 SPS: a burst's LDS slice is 4.4 KB and the kernel fits 128 registers, so 16 waves share a CU as in the 4-SPS kernel;
//  the generic 4-SPS instantiations keep 12 waves and their 147-168 registers)
#define TRX_WPB_OF(SPS_) ((SPS_) == 1 ? 16 : TRX_WPB)
template <int SPS, bool CF32, int NLD>
__global__ void __launch_bounds__(TRX_WPB_OF(SPS) * WAVE, (SPS == 1 ? 4 : 3))
burst_pull_kernel(const void *__restrict__ iq_, const trxhip_burst_params *__restrict__ params,
		  trxhip_burst_result *__restrict__ results, float *__restrict__ soft,
		  const trx_tables *__restrict__ tab, const float4 *__restrict__ ebp_in,
		  unsigned n_bursts, int L, float thresh, float full_scale, int soft_stride, int slice)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = uni((int)(threadIdx.x >> 6));                  // wave-uniform: burst index and its addresses live in SGPRs
	const int waves_per_block = blockDim.x >> 6;

	// ---- LDS carve: [tables][per-wave slices]
	float *sincv = reinterpret_cast<float *>(smem);                 // [4128] swizzled sinc LUT
	float *dfilt = sincv + TRX_SINCV_LDS;                          // [64][20] fractional-delay filters
	c32 *rrot = reinterpret_cast<c32 *>(dfilt + TRX_DELAY_FILTS * TRX_DELAY_HLEN);   // [160] reverse rotation
	float *gdec = reinterpret_cast<float *>(rrot + 160);           // [16] decimator taps
	c32 *lseq = reinterpret_cast<c32 *>(gdec + 16);                // [376] training sequences
	float *lhdr = reinterpret_cast<float *>(lseq + LSEQ_TAPS);     // [20][8] sequence headers
	const int xs_len = TRX_PAD + L + TRX_PAD;
	const int xs_alloc = (xs_len + 1) & ~1;
	const int slice_c32 = xs_alloc + TRX_DEC_LEN + TRX_CZ_LEN;
	c32 *wbase = reinterpret_cast<c32 *>(smem + TRX_TABLES_LDS_BYTES) + (size_t)wave * slice_c32;
	int *const wg_next = reinterpret_cast<int *>(reinterpret_cast<c32 *>(smem + TRX_TABLES_LDS_BYTES) + (size_t)waves_per_block * slice_c32);   // work counter
	c32 *const xs = wbase + TRX_PAD;                               // burst sample 0
	c32 *const dec = wbase + xs_alloc;                             // 1-SPS (decimated) burst, zero tail
	c32 *const cz = dec + TRX_DEC_LEN + TRX_CZ_PAD;                // zero-padded correlation

	// ---- one-time staging (workgroup-wide) of every table; zero this wave's slice (pads stay zero)
	for (int i = threadIdx.x; i < TRX_SINCV_LDS; i += blockDim.x)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	for (int i = threadIdx.x; i < TRX_DELAY_FILTS * TRX_DELAY_HLEN; i += blockDim.x)
		dfilt[i] = (&tab->delay_filt[0][0])[i];
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
		// header of LDS sequence slot: slots 0-7 TSC, 8-10 RACH, 11-18 EDGE, 19 dummy
		const int slot = i / 8;
		const int s = (slot < 8) ? TRX_SEQ_TSC0 + slot : (slot < 11) ? TRX_SEQ_RACH0 + (slot - 8) : (slot < 19) ? TRX_SEQ_EDGE0 + (slot - 11) : TRX_SEQ_DUMMY;
		lhdr[i] = reinterpret_cast<const float *>(&tab->seq[s].gain)[i % 8];
	}
	for (int i = lane; i < slice_c32; i += WAVE)
		wbase[i] = make_float2(0.0f, 0.0f);
	if (threadIdx.x == 0)
		*wg_next = waves_per_block;
	__syncthreads();

	const float fs_db = 6.02059991f * __log2f(full_scale);          // 20*log10(full_scale)
	const PeakConst pkc = peak_const(threadIdx.x & (WAVE - 1));      // lane constants of the TOA bisection
	// the workgroup's bursts are blockIdx.x, blockIdx.x + gridDim.x, ...; its waves claim them one ahead from an LDS counter
	// (a static split leaves the CU under-occupied for the last third of the kernel, see burst_pull4_kernel)
	const unsigned n_wg = gridDim.x;
	// items are handed out in groups of 16 CONSECUTIVE bursts (group g belongs to workgroup g % gridDim.x): neighbouring
	// bursts share the 128-byte line their boundary falls in, and with them on one CU that line is fetched from HBM once
	const unsigned n_groups = (n_bursts + 15u) >> 4;
	const unsigned my_groups = (blockIdx.x < n_groups) ? (n_groups - blockIdx.x + n_wg - 1) / n_wg : 0u;
	unsigned items = my_groups << 4;
	if (my_groups && (my_groups - 1) * n_wg + blockIdx.x == n_groups - 1)
		items -= (n_groups << 4) - n_bursts;                        // the batch's last group may be short
	auto burst_of = [&](unsigned jj) { return (((jj >> 4) * n_wg + blockIdx.x) << 4) + (jj & 15u); };

	// Software prefetch: the raw samples and the parameter word of this wave's NEXT burst sit in
	// registers (NLD dwords per lane, coalesced 256 B per wave-load) while the current burst is
	// processed, so the HBM latency is hidden behind compute.  NLD == 0: generic burst lengths, plain loop.
	uint32_t pre_i[NLD > 0 ? NLD : 1];
	c32 pre_c[(NLD > 0 && CF32) ? NLD : 1];
	uint32_t pre_prm = 0u;                                         // {type u8, tsc u8, max_toa u16}
	auto prefetch = [&](unsigned bb) {
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
				pre_i[r] = (r < NLD - 1 || i < L) ? src[i] : 0u;   // host guarantees L > 64*(NLD-1)
			}
		}
	};
	if ((unsigned)wave < items)
		prefetch(burst_of((unsigned)wave));

	DIAG_DECL;
#ifdef TRX_WHATIF_PAIR
	WhatIf wi_none = { 0, 0 };
#define WI_1SPS , wi_none
#else
#define WI_1SPS
#endif
	unsigned j_next = 0;
	for (unsigned j = (unsigned)wave; j < items; j = j_next) {
		const unsigned b = burst_of(j);
		const int ticket = claim_issue(wg_next);                   // this wave's next item; taken at prefetch time below
		const unsigned prm0 = (unsigned)uni((int)pre_prm);
		const int type = prm0 & 0xff;
		const int tsc = (prm0 >> 8) & 0xff;
		const int max_toa = prm0 >> 16;

		int rc = 0;
		float toa = 0.0f, ci = 0.0f, energy = 0.0f, rssi = 0.0f;
		c32 amp = make_float2(0.0f, 0.0f);
		int out_tsc = 0, clip = 0, idle = 1, nbits = 0;
		float *so = soft ? soft + (size_t)b * soft_stride : nullptr;

		// ---- phase 0: registers -> fp32 LDS (convert_short_float, arch/common/convert_base.c:27-31, fused
		// into the load); clip scan (maxAmplitude) and energyDetect partial sums ride on the same values
		float amax = 0.0f, epart = 0.0f;
		int win = 20 * SPS;                                         // energyDetect window (:725), stride 4 (:1582)
		if (win > L) win = L;
		if (NLD > 0) {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				if (r < NLD - 1 || i < L) {
					c32 v;
					if (CF32) v = pre_c[r];
					else v = make_float2((float)(int16_t)(pre_i[r] & 0xffffu), (float)(int16_t)(pre_i[r] >> 16));
					xs[i] = v;
					asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(v.x), "v"(v.y));   // maxAmplitude(): one instruction per sample
					if (r * WAVE < 4 * 20 * SPS)                    // compile-time: rounds that can hold samples 4i, i < win
						if ((lane & 3) == 0 && i < 4 * win)
							epart += norm2(v);
				}
			}
			j_next = (unsigned)claim_take(ticket);
			if (j_next < items)
				prefetch(burst_of(j_next));
		} else {
			j_next = (unsigned)claim_take(ticket);
			if (j_next < items)
				pre_prm = reinterpret_cast<const uint32_t *>(params)[2 * (size_t)burst_of(j_next)];
			for (int i = lane; i < L; i += WAVE) {
				c32 v;
				if (CF32) {
					v = (reinterpret_cast<const c32 *>(iq_) + (size_t)b * L)[i];
				} else {
					const uint32_t u = (reinterpret_cast<const uint32_t *>(iq_) + (size_t)b * L)[i];
					v = make_float2((float)(int16_t)(u & 0xffffu), (float)(int16_t)(u >> 16));
				}
				xs[i] = v;
				amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
				if ((i & 3) == 0 && i < 4 * win)
					epart += norm2(v);
			}
		}

		if (type != TRXHIP_OFF) {                                   // Transceiver.cpp:704-707
			clip = __ballot(amax > TRX_CLIP_THRESH) != 0ull;        // maxAmplitude() > 30000 (:1711-1722, :1746): some lane saw a larger component
			// energyDetect(burst, 20*sps) (:1573-1585), tree-summed; RSSI (Transceiver.cpp:741,751) in fp32
			energy = wave_sum(epart) / (float)win;
			if (!ABL(2))
				rssi = fs_db - 3.01029996f * __log2f(energy);       // 20*log10(fs/sqrt(e)), Transceiver.cpp:741,751
			wave_sync();

			if (ebp_in) {
				// demodAnyBurst() on its own (sigProcLib.h:151-152): the caller supplies type, toa and amp
				const float4 e = ebp_in[b];
				rc = (type == TRXHIP_OFF || type == TRXHIP_IDLE) ? 0 : type;
				toa = unif(e.x);
				amp = make_float2(unif(e.y), unif(e.z));
				out_tsc = tsc;
			} else if ((type != TRXHIP_IDLE || (slice & TRXHIP_FLAG_IDLE_DUMMY)) && !ABL(3)) {   // Transceiver.cpp:754-755
				// ---- detectAnyBurst (:1926-1957)
				DetectOut d;
				if (SPS == 4) {
					// downsampleBurst (:1587-1601) restricted to what correlate/computeCI read:
					// dec[i] = sum_k xs[4i-15+k] * g[k], i in [lo, hi)
					auto decimate = [&](int lo, int hi) {
						for (int i = lo + lane; i < hi; i += WAVE) {
							const c32 *xp = xs + 4 * i - 15;
							float yr = 0.0f, yi = 0.0f;
#pragma unroll
							for (int k = 0; k < 16; k++) {
								const c32 x = xp[k];
								const float g = gdec[k];
								yr += x.x * g;
								yi += x.y * g;
							}
							dec[i] = make_float2(yr, yi);
						}
						wave_sync();
					};
					rc = detect_any_burst<true, false>(type, tsc, max_toa, clip, decimate, dec, 156, cz, lseq, lhdr, thresh, sincv,
								    pkc, lane, slice, 1 /* multiplying correlation */, &d DIAG_PASS);
				} else if (type == TRXHIP_TSC && tsc < 8 && max_toa <= 33 && L >= 148 && !(slice & TRX_IFLAG_NO_UNIT)) {
					// Round 4: the common slot at 1 SPS, straight-line, as in burst_pull4_kernel.  A normal burst is ONE
					// detectGeneralBurst() window (analyzeTrafficBurst, :1887-1904: start 71, len 16 + max_toa <= 49) over the
					// burst itself (no decimation at 1 SPS); its 31 + max_toa samples xs[56 ..] lie inside the burst and the
					// 20 zero samples either side of it cover every padded read, so the correlation is the addition-only
					// corr_unit() form (same bits, guard below) with lane = lag in one round -- where the generic dispatch
					// multiplies every tap, range-checks every read and walks the candidate loop (~170 scalar, ~100 vector
					// instructions per burst more).
					const int len = 16 + max_toa;
					const int unit_bad = (__ballot(unit_unsafe(xs[56 + lane]) && lane < 15 + len) != 0ull) ? 1 : 0;
					const float *const hdr = lhdr + 8 * tsc;
					const int hit = detect_burst_h<true, false, false>(xs, L, cz, lseq + LSEQ_TSC(tsc), hdr, 16, thresh, 71, len, sincv, pkc, lane,
										    &d.toa, &d.amp, &d.ci, NoToaHook(), nullptr, slice, unit_bad ? -1 : tsc DIAG_PASS WI_1SPS);
					wave_sync();
					rc = hit ? TRXHIP_TSC : (clip ? -TRXHIP_SIGERR_CLIP : 0);                      // :1764, :1953-1954
					d.toa -= 10.0f;                                                                 // :1768
					d.tsc = tsc;
				} else {
					auto nothing = [](int, int) {};
					rc = detect_any_burst<false, false>(type, tsc, max_toa, clip, nothing, xs, L, cz, lseq, lhdr, thresh, sincv,
								     pkc, lane, slice, 1, &d DIAG_PASS);
				}
				if (rc > 0) { toa = d.toa; amp = d.amp; ci = d.ci; out_tsc = d.tsc; }
			}
		}

		// ---- demodAnyBurst -> demodGmskBurst (:2055-2072) ----
		if (rc > 0 && !ABL(0)) {
			// demodCommon (:2030-2048): delayVector(burst, -toa*sps), scaleVector(1/amp)
			const float delay = -toa * (float)SPS;
			const int whole = (int)floorf(delay);
			const float frac = delay - (float)whole;
			const bool use_filt = ((double)fabsf(frac) > 1e-2) && !ABL(4);  // :1056
			const int fidx = use_filt ? (int)floorf(frac * (float)TRX_DELAY_FILTS) : 0;   // :1057
			const float4 *hf4 = reinterpret_cast<const float4 *>(dfilt + uni(fidx) * TRX_DELAY_HLEN);
			// (complex) 1.0 / amp = (1,0) * amp.inv()   (Complex.h:75,144-150)
			const float an = norm2(amp);
			const c32 ainv = make_float2(amp.x / an, -amp.y / an);
			const c32 scale = cmul(make_float2(1.0f, 0.0f), ainv);

			const int n_out = (SPS == 4) ? 624 : L;                         // samples the next stage reads
			constexpr int R = (SPS == 4) ? 10 : 3;                          // outputs per lane
			c32 yv[R];
			{
				const int n0 = lane * R;
				const int m0 = n0 - whole;                                  // y[n] = fshift[n - whole]
				int mc = m0;
				if (mc < -10) mc = -10;
				if (mc > L) mc = L;
				const c32 *xp = xs + mc - 9;
				if (use_filt) {
					// fshift[m] = sum_k X(m - 9 + k) * h[k]  (convolve NO_DELAY, 20 real taps; :1060)
					// tap-outer / output-inner: each output still accumulates k = 0..19 in order, but only a
					// sliding window of R samples (+ R accumulators) is live instead of all R+19 inputs
					trx_v2f hp[TRX_DELAY_HLEN / 2];                             // taps 2q, 2q + 1 in one register pair
#pragma unroll
					for (int q = 0; q < TRX_DELAY_HLEN / 4; q++) {          // 5 LDS broadcast reads
						const float4 h4 = hf4[q];
						hp[2 * q + 0] = (trx_v2f){ h4.x, h4.y };
						hp[2 * q + 1] = (trx_v2f){ h4.z, h4.w };
					}
					constexpr int D = 3;                                    // LDS read-ahead, in taps
					// Taps 0, 17, 18, 19 are exactly 0.0f in all 64 filters (the sinc LUT is zero beyond 8 pi, sigProcLib.cpp:990-998;
					// tests/test_capi_cpu.py): fl(x * 0) = +-0 and y + (+-0) == y for every finite x (y starts at +0 and can never
					// become -0), so those four steps of the reference's loop change nothing and are skipped -- 16 multiply-adds per
					// output instead of 20, same bits (as in burst_pull4_kernel's exact demodulator)
					constexpr int K0 = 1, K1 = 17;
					// As there, in explicit packed instructions: one v_pk_mul_f32 (the tap picked from its pair by op_sel) and one
					// v_pk_add_f32 per step -- product, then sum, k ascending: the reference's two roundings per step.  (From the
					// scalar form the compiler builds unpacked multiplies and adds with moves in between.)
					trx_v2f xr[R + 19];
					trx_v2f ya[R];
#pragma unroll
					for (int j = 0; j < R; j++)
						ya[j] = (trx_v2f){ 0.0f, 0.0f };
#pragma unroll
					for (int j = K0; j < K0 + R - 1 + D; j++) {
						const c32 t = xp[j];
						xr[j] = (trx_v2f){ t.x, t.y };
					}
#pragma unroll
					for (int k = K0; k < K1; k++) {
						const trx_v2f hpair = hp[k >> 1];
						if (R - 1 + D + k < R + K1 - 1) {
							const c32 t = xp[R - 1 + D + k];
							xr[R - 1 + D + k] = (trx_v2f){ t.x, t.y };
						}
						trx_v2f pr[R];
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
						__builtin_amdgcn_sched_barrier(0);              // keep the window short: no load hoisting
					}
#pragma unroll
					for (int j = 0; j < R; j++)
						yv[j] = make_float2(ya[j].x, ya[j].y);
				} else {
#pragma unroll
					for (int j = 0; j < R; j++)
						yv[j] = xp[9 + j];
				}
#pragma unroll
				for (int j = 0; j < R; j++) {
					const int m = m0 + j;
					const bool ok = (mc == m0) && ((unsigned)m < (unsigned)L);
					const c32 v = ok ? yv[j] : make_float2(0.0f, 0.0f);
					yv[j] = cmul(v, scale);                                 // scaleVector (:1198-1205)
				}
			}
			wave_sync();                 // every lane has read its inputs: safe to overwrite in place
			{
				const int n0 = lane * R;
#pragma unroll
				for (int j = 0; j < R; j++)
					if (n0 + j < n_out)
						xs[n0 + j] = yv[j];
			}
			wave_sync();

			// downsampleBurst (4 SPS) + GMSKReverseRotate + real part + vectorSlicer
			const int nsoft = (SPS == 4) ? 156 : L;
			idle = 0;
			if (rc == TRXHIP_EDGE) {
				// demodEdgeBurst (:2105-2128): decimate everything into LDS, then equalise / derotate / slice
				if (SPS == 4) {
					for (int i = lane; i < 156; i += WAVE) {
						const c32 *xp = xs + 4 * i - 15;
						float yr = 0.0f, yi = 0.0f;
#pragma unroll
						for (int k = 0; k < 16; k++) {
							const c32 x = xp[k];
							const float g = gdec[k];
							yr += x.x * g;
							yi += x.y * g;
						}
						dec[i] = make_float2(yr, yi);
					}
					wave_sync();
					ci = edge_post(dec, 156, tab, so, soft_stride, slice, lane);
				} else {
					ci = edge_post(xs, L, tab, so, soft_stride, slice, lane);
				}
				nbits = 444;
			} else {
			nbits = 148;
			if (so) {
				const int nwrite = (slice & 1) ? nbits : nsoft;
				for (int i = lane; i < soft_stride; i += WAVE) {
					float sv = 0.0f;
					if (i < nwrite && !ABL(5)) {
						c32 d;
						if (SPS == 4) {
							const c32 *xp = xs + 4 * i - 15;
							float yr = 0.0f, yi = 0.0f;
#pragma unroll
							for (int k = 0; k < 16; k++) {
								const c32 x = xp[k];
								const float g = gdec[k];
								yr += x.x * g;
								yi += x.y * g;
							}
							d = make_float2(yr, yi);
						} else {
							d = xs[i];
						}
						const c32 r = rrot[i];
						sv = r.x * d.x - r.y * d.y;                         // real(rot * x)  (:2066-2068)
						if (slice & 1)                                      // vectorSlicer (:546-556): clamp(0.5*(s+1), 0, 1)
							sv = __builtin_amdgcn_fmed3f(fmaf(0.5f, sv, 0.5f), 0.0f, 1.0f);   // 0.5 * (x + 1), bit for bit
					}
					so[i] = sv;
				}
			}
			}
			wave_sync();
			// the in-place delayed burst leaves sample L-1 (and nothing else) stale: harmless, the next
			// burst overwrites [0, L) completely
		} else {
			if (so)
				for (int i = lane; i < soft_stride; i += WAVE)
					so[i] = 0.0f;
		}

		// ---- result record: 32 bytes, one dword per lane 0..7
		if (lane < 8) {
			const bool det = rc > 0;
			uint32_t word = (uint32_t)rc;
			word = (lane == 1) ? __float_as_uint(det ? toa : 0.0f) : word;
			word = (lane == 2) ? __float_as_uint(det ? amp.x : 0.0f) : word;
			word = (lane == 3) ? __float_as_uint(det ? amp.y : 0.0f) : word;
			word = (lane == 4) ? __float_as_uint(det ? ci : 0.0f) : word;
			word = (lane == 5) ? __float_as_uint(energy) : word;
			word = (lane == 6) ? __float_as_uint(rssi) : word;
			word = (lane == 7) ? ((uint32_t)(det ? out_tsc : 0) | ((uint32_t)clip << 8) | ((uint32_t)idle << 16) |
					      ((uint32_t)(nbits / 4) << 24)) : word;
			reinterpret_cast<uint32_t *>(results + b)[lane] = word;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from trx_capi.cpp)
// ------------------------------------------------------------------------------------------------
extern "C" size_t trx_pull_lds_bytes(int L, int waves_per_block)
{
	const int xs_len = TRX_PAD + L + TRX_PAD;
	const size_t slice_c32 = ((xs_len + 1) & ~1) + TRX_DEC_LEN + TRX_CZ_LEN;
	return TRX_TABLES_LDS_BYTES + (size_t)waves_per_block * slice_c32 * sizeof(c32) + 16;   // + the workgroup's work counter
}

extern "C" int trx_launch_pull4(unsigned *d_pool_ctr, const void *d_iq, int cf32, const trxhip_burst_params *d_params,
				trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, const float *d_ebp_in,
				size_t n_bursts, int L, float thresh, float full_scale, int soft_stride, int flags, int n_cu,
				hipStream_t stream);

extern "C" int trx_launch_pull(unsigned *d_pool_ctr, const void *d_iq, int cf32, const trxhip_burst_params *d_params,
			       trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab, const float *d_ebp_in,
			       size_t n_bursts, int L, int sps, float thresh, float full_scale,
			       int soft_stride, int slice, int n_cu, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	// the transceiver's 4-SPS burst size gets the production kernel (polyphase LDS layout, fused or exact demod)
	if (sps == 4 && L >= 624 && L <= 628)
		return trx_launch_pull4(d_pool_ctr, d_iq, cf32, d_params, d_results, d_soft, d_tab, d_ebp_in, n_bursts, L, thresh, full_scale,
					soft_stride, slice, n_cu, stream);
	// as many waves per workgroup as the 160 KB of LDS admit (12 at L = 625), one workgroup per CU
	int wpb = TRX_WPB_OF(sps);
	while (wpb > 1 && trx_pull_lds_bytes(L, wpb) > 160 * 1024)
		wpb--;
	const size_t lds = trx_pull_lds_bytes(L, wpb);
	if (lds > 160 * 1024)
		return TRXHIP_EINVAL;
	size_t need = (n_bursts + 15) / 16;                             // work is handed out in groups of 16 bursts
	size_t grid = (size_t)n_cu * (size_t)((160 * 1024) / lds);
	if (grid > need) grid = need;

#define LAUNCH(SPS_, CF_, NLD_)                                                                                 \
	do {                                                                                                    \
		auto k = burst_pull_kernel<SPS_, CF_, NLD_>;                                                    \
		TRX_ARM_DYNAMIC_LDS(k);                                                                         \
		hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(wpb * WAVE), lds, stream, d_iq, d_params, d_results, \
				   d_soft, d_tab, reinterpret_cast<const float4 *>(d_ebp_in), (unsigned)n_bursts, L, thresh,     \
				   full_scale, soft_stride, slice);                                             \
	} while (0)

	// NLD = dword loads per lane held in registers for the prefetched burst (0 = generic length, no prefetch)
	if (sps == 4) {
		if (L <= 640) { if (cf32) LAUNCH(4, true, 10); else LAUNCH(4, false, 10); }
		else          { if (cf32) LAUNCH(4, true, 0);  else LAUNCH(4, false, 0); }
	} else {
		if (cf32) LAUNCH(1, true, 3); else LAUNCH(1, false, 3);
	}
#undef LAUNCH
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
