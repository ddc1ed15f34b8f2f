// trx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for osmo-trx's receive-side burst DSP.
//
// Hot kernel: burst_pull_kernel<SPS> = the DSP core of Transceiver::pullRadioVector()
// (Transceiver52M/Transceiver.cpp:724-803): convert_short_float -> energyDetect -> clip check ->
// detectAnyBurst -> demodAnyBurst -> vectorSlicer, for a batch of independent bursts.
//
// Mapping (MI355X-first, not a translation of the SSE code):
//   * ONE WAVEFRONT (64 lanes) PER BURST, persistent waves grid-striding over the batch; a burst
//     (625 x int16 IQ = 2500 B) is read from HBM exactly once with coalesced dword loads, converted
//     to fp32 in flight and kept in that wave's private LDS slice (~7.6 KB) until its soft bits and
//     32-byte result record are written: zero intermediate HBM traffic.
//   * waves never synchronise with each other after the one-time staging of the sinc-interpolation
//     table into LDS; intra-wave ordering relies on wave-lockstep LDS execution (wave_sync()).
//   * FIR phases are register-blocked (fractional-delay: 10 outputs x 20 taps per lane from 29 LDS
//     reads); wave-uniform taps/training sequences come in through the scalar cache (SGPR operands),
//     per-lane gathers (sinc LUT) from a bank-swizzled LDS table.
//   * peak/TOA: argmax by __shfl_xor butterfly; the reference's 9-step early/late bisection
//     (19 data-dependent sinc interpolations) is evaluated SPECULATIVELY: the binary decision tree is
//     expanded across lanes (2 rounds: levels 0-4, then 5-8 + the 16 possible final positions), each
//     lane doing one sequential 21-tap interpolation, then the path is walked with lane shuffles.
//   * no MFMA: these are short 1-D real/complex convolutions (<= 40 taps), HBM/VALU/LDS work.
//
// Numerics: every sum that feeds a DECISION (correlation, peak ratio, bisection compares, filter
// choice) is accumulated in the reference's generic-C order (arch/common/convolve_base.c:28-54) and
// the file is compiled with -ffp-contract=off, so rc / TOA / amp / soft bits are bit-identical to the
// generic-C reference.  Only energyDetect (tree-summed, <=1e-6 rel), log2f (C/I) and log10 (RSSI)
// differ at the last-ulp level.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "trx_tables.h"
#include "../../include/trxhip.h"

#define WAVE 64
#define TRX_PAD 20                 // zero samples kept on both sides of a burst in LDS
#define TRX_DEC_LEN 160            // decimated burst (156 used)
#define TRX_CORR_MAX 128           // head + tail <= 16 + TRXHIP_MAX_TOA
#define TRX_CZ_PAD 12              // zero samples either side of the correlation (interpolatePoint reach)
#define TRX_CZ_LEN (TRX_CZ_PAD + TRX_CORR_MAX + TRX_CZ_PAD)
#define TRX_SINCV_LDS (TRX_SINCV_LEN + 32)   // + zero tail: q = 4096 is addressed when the fraction is 0
#define TRX_CLIP_THRESH 30000.0f   // sigProcLib.cpp:49
#define TRX_WPB 8                  // waves (= bursts in flight) per workgroup

typedef float2 c32;

// ------------------------------------------------------------------------------------------------
// small helpers
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
// value held by lane `l` (l wave-uniform): v_readlane_b32, no LDS round trip
__device__ __forceinline__ float lane_val(float v, int l)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l)));
}
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Complex.h:113 norm2(): i*i + r*r
__device__ __forceinline__ float norm2(c32 v) { return v.y * v.y + v.x * v.x; }
// Complex.h:74 operator*(Complex)
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v = fmaxf(v, __shfl_xor(v, o, WAVE));
	return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, WAVE);
	return v;
}

// per-wave LDS carve
struct WaveLds {
	c32 *xs;      // [TRX_PAD + L + TRX_PAD]   burst, later overwritten in place by the delayed+scaled burst
	c32 *dec;     // [TRX_DEC_LEN]             1-SPS burst (decimated) ; zero beyond 156
	c32 *corr;    // [TRX_CZ_LEN]  zero-padded correlation
};

struct SeqWin {   // one detectGeneralBurst() call: sequence + window (sigProcLib.cpp:1732-1771)
	int seq;      // index into tables->seq
	int target, head, tail;
};

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
__device__ __forceinline__ c32 interp_point(const c32 *cz, int ix512, const float *sincv)
{
	const int fl = ix512 >> 9;                       // floor(ix)
	const int f = ix512 & 511;                       // fractional part * 512
	const int fs = trx_sincv_swz(f);                 // taps i <= fl : q = 512*(fl-i) + f
	const int g = 512 - f;                           // taps i >  fl : q = 512*(i-fl-1) + (512 - f)
	const int gs = (f == 0) ? 512 : trx_sincv_swz(g);
	const c32 *c = cz + (fl - 7);
	const float *sa = sincv + fs;
	const float *sb = sincv + gs;
	c32 p = make_float2(0.0f, 0.0f);
#pragma unroll
	for (int u = 0; u < 8; u++) {                    // i = fl-7 .. fl   (k = 7 .. 0)
		const c32 v = c[u];
		const float w = sa[512 * (7 - u)];
		p.x += v.x * w;
		p.y += v.y * w;
	}
#pragma unroll
	for (int u = 0; u < 8; u++) {                    // i = fl+1 .. fl+8 (k = 0 .. 7)
		const c32 v = c[8 + u];
		const float w = sb[512 * u];
		p.x += v.x * w;
		p.y += v.y * w;
	}
	return p;
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

// ------------------------------------------------------------------------------------------------
// peakDetect() (sigProcLib.cpp:1141-1186) with the early/late bisection expanded across lanes.
// All lanes return the same (toa512, value).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void peak_detect_spec(const c32 *cz, int max_idx, const float *sincv,
						  int lane, int *toa512_out, c32 *val_out)
{
	int E = (max_idx - 1) * 512;                     // earlyIndex * 512
	bool tie = false;

	// ---- round A: levels 0..4 (incr = 1/2 .. 1/32): heap node n on lanes 2n (early) and 2n+1 (late)
	{
		const int ix = E + node_offset(lane >> 1, 256) + ((lane & 1) ? 1024 : 0);
		float nv = 0.0f;
		if (lane < 62)
			nv = norm2(interp_point(cz, ix, sincv));
		int node = 0;
#pragma unroll
		for (int Lw = 0; Lw < 5; Lw++) {
			const float ne = lane_val(nv, 2 * node);
			const float nl = lane_val(nv, 2 * node + 1);
			if (!tie) {
				if (ne < nl)      { E += (256 >> Lw); node = 2 * node + 2; }
				else if (ne > nl) { E -= (256 >> Lw); node = 2 * node + 1; }
				else tie = true;                      // "else break", :1170
			}
		}
	}
	int final_ix;
	c32 val;
	if (!tie) {
		// ---- round B: levels 5..8 (incr = 1/64 .. 1/512) on lanes 0..29, the 16 possible final
		// positions (earlyIndex + 1) on lanes 32..47
		int ix;
		if (lane < 32)
			ix = E + node_offset(lane >> 1, 8) + ((lane & 1) ? 1024 : 0);
		else
			ix = E + (2 * (lane & 15) - 15) + 512;
		c32 pv = make_float2(0.0f, 0.0f);
		if (lane < 30 || (lane >= 32 && lane < 48))
			pv = interp_point(cz, ix, sincv);
		const float nv = norm2(pv);
		int node = 0, offB = 0;
#pragma unroll
		for (int Lw = 0; Lw < 4; Lw++) {
			const float ne = lane_val(nv, 2 * node);
			const float nl = lane_val(nv, 2 * node + 1);
			if (!tie) {
				if (ne < nl)      { offB += (8 >> Lw); node = 2 * node + 2; }
				else if (ne > nl) { offB -= (8 >> Lw); node = 2 * node + 1; }
				else tie = true;
			}
		}
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
// detectBurst() on the 1-SPS signal `sig[0..sig_len)` (sigProcLib.cpp:1649-1709), corr in LDS.
// Returns rc (1 / 0); on 1 fills toa (symbols, before "- head"), amp, ci.  Wave-uniform.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int detect_burst(const c32 *sig, int sig_len, c32 *cz, const trx_seq *__restrict__ sq,
					     float thresh, int start, int len, const float *sincv, int lane,
					     float *toa_out, c32 *amp_out, float *ci_out)
{
	const int N = sq->n;

	// ---- correlate: corr[i] = sum_k SIG(i + start - (N-1) + k) * seq[k]   (:1674, convolve_base.c:72-85)
	// N is 16 (TSC/EDGE/dummy), 40 (RACH) or 64 (SCH): tap loop unrolled by 8 so the LDS reads pipeline
	for (int i = lane; i < len; i += WAVE) {
		float yr = 0.0f, yi = 0.0f;
		const int base = i + start - (N - 1);
		for (int k0 = 0; k0 < N; k0 += 8) {
			c32 x[8];
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const int j = base + k0 + u;
				x[u] = (j >= 0 && j < sig_len) ? sig[j] : make_float2(0.0f, 0.0f);
			}
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const float hr = sq->taps[k0 + u].re, hi = sq->taps[k0 + u].im;   // wave-uniform -> scalar loads
				yr += x[u].x * hr - x[u].y * hi;
				yi += x[u].x * hi + x[u].y * hr;
			}
		}
		cz[i] = make_float2(yr, yi);
	}
	if (lane < TRX_CZ_PAD)
		cz[len + lane] = make_float2(0.0f, 0.0f);          // right zero pad (len varies per burst)
	wave_sync();

	// ---- fastPeakDetect (:1120-1139): first strict maximum of |corr|^2
	float best = 0.0f;
	int bidx = -1;
	for (int i = lane; i < len; i += WAVE) {
		const float v = norm2(cz[i]);
		if (v > best) { best = v; bidx = i; }
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const float ov = __shfl_xor(best, o, WAVE);
		const int oi = __shfl_xor(bidx, o, WAVE);
		if (ov > best || (ov == best && oi >= 0 && (bidx < 0 || oi < bidx))) { best = ov; bidx = oi; }
	}
	bidx = uni(bidx);
	if (bidx < 0)
		return 0;                                    // toa = -1 < 3
	const float toa0 = (float)bidx;
	if ((toa0 < 3.0f) || (toa0 > (float)(len - 3)))   // :1683
		return 0;
	const c32 amp0 = cz[bidx];

	// ---- computePeakRatio (:1541-1571), sequential sum (it gates a decision)
	{
		int num = 0;
		float avg = 0.0f;
		const int peak = bidx;                       // rint(toa) of an integer
#pragma unroll
		for (int i = 2; i <= 5; i++) {
			if (peak - i >= 0)  { avg += norm2(cz[peak - i]); num++; }
			if (peak + i < len) { avg += norm2(cz[peak + i]); num++; }
		}
		if (num < 5)
			return 0;
		const float rms = (float)((double)sqrtf(avg / (float)num) + 0.00001);
		const float ratio = sqrtf(norm2(amp0)) / rms;
		if (ratio < thresh)
			return 0;
	}

	// ---- peakDetect (:1695): refined TOA (multiple of 1/512) and interpolated correlation value
	int toa512;
	c32 xcorr;
	// interpolatePoint() never reads the last correlation sample (:1105, :1109): zero it in the padded copy
	wave_sync();
	if (lane == 0)
		cz[len - 1] = make_float2(0.0f, 0.0f);
	wave_sync();
	peak_detect_spec(cz, bidx, sincv, lane, &toa512, &xcorr);
	toa512 = uni(toa512);
	xcorr.x = unif(xcorr.x);
	xcorr.y = unif(xcorr.y);
	const float toa = (float)toa512 * (1.0f / 512.0f);   // exact

	// ---- computeCI (:1608-1639)
	float ci = 0.0f;
	{
		// roundf(toa): toa is k/512 -> round half away from zero on integers
		const int rt = (toa512 >= 0) ? ((toa512 + 256) >> 9) : -((-toa512 + 256) >> 9);
		const int ps = start + 1 - N + rt;
		if (ps >= 0 && ps + N <= sig_len) {
			// S = sum_i |sig[ps+i]|^2 in index order: lane i squares one sample, the sum walks the lanes
			const float pw = norm2(sig[ps + (lane < N ? lane : 0)]);
			float S = 0.0f;
			for (int i = 0; i < N; i++)
				S += lane_val(pw, i);
			S /= (float)N;
			const float C = norm2(xcorr) / sq->ci_den;
			ci = 3.0103f * log2f(C / (S - C));
		}
	}

	*amp_out = cmul(xcorr, make_float2(sq->gain_inv.re, sq->gain_inv.im));   // xcorr / sync->gain  (:1701)
	*toa_out = toa - sq->toa;                                              // :1704
	*ci_out = ci;
	return 1;
}

// ------------------------------------------------------------------------------------------------
// the hot kernel
// ------------------------------------------------------------------------------------------------
template <int SPS, bool CF32, int NLD, int WPB>
__global__ void __launch_bounds__(WPB * WAVE, (WPB == 8) ? 4 : 3)
burst_pull_kernel(const void *__restrict__ iq_, const trxhip_burst_params *__restrict__ params,
		  trxhip_burst_result *__restrict__ results, float *__restrict__ soft,
		  const trx_tables *__restrict__ tab,
		  unsigned n_bursts, int L, float thresh, float full_scale, int soft_stride, int slice)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = threadIdx.x >> 6;
	const int waves_per_block = blockDim.x >> 6;

	// ---- LDS carve: [sincv 16 KB | per-wave slices]
	float *sincv = reinterpret_cast<float *>(smem);
	const int xs_len = TRX_PAD + L + TRX_PAD;
	const int slice_c32 = ((xs_len + 1) & ~1) + TRX_DEC_LEN + TRX_CZ_LEN;
	c32 *wbase = reinterpret_cast<c32 *>(smem + TRX_SINCV_LDS * sizeof(float)) + (size_t)wave * slice_c32;
	WaveLds w;
	w.xs = wbase;
	w.dec = wbase + ((xs_len + 1) & ~1);
	w.corr = w.dec + TRX_DEC_LEN;

	// one-time staging (block-wide): sinc LUT; zero this wave's pads
	for (int i = threadIdx.x; i < TRX_SINCV_LDS; i += blockDim.x)
		sincv[i] = (i < TRX_SINCV_LEN) ? tab->sincv[i] : 0.0f;
	for (int i = lane; i < slice_c32; i += WAVE)
		wbase[i] = make_float2(0.0f, 0.0f);
	__syncthreads();

	const unsigned total_waves = gridDim.x * waves_per_block;
	const unsigned first = blockIdx.x * waves_per_block + wave;

	// Software prefetch: the raw samples of this wave's NEXT burst sit in registers (NLD dwords per lane,
	// coalesced 256 B per wave-load) while the current burst is being processed, so the ~1-2 us HBM
	// latency is hidden behind compute instead of being paid serially per burst.  NLD == 0: generic
	// burst lengths, plain loop.
	uint32_t pre_i[NLD > 0 ? NLD : 1];
	c32 pre_c[(NLD > 0 && CF32) ? NLD : 1];
	auto prefetch = [&](unsigned bb) {
		if (CF32) {
			const c32 *src = reinterpret_cast<const c32 *>(iq_) + (size_t)bb * L;
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				pre_c[r] = (i < L) ? src[i] : make_float2(0.0f, 0.0f);
			}
		} else {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(iq_) + (size_t)bb * L;
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				pre_i[r] = (i < L) ? src[i] : 0u;
			}
		}
	};
	if (NLD > 0 && first < n_bursts)
		prefetch(first);

	const int lane_outer = lane;
	for (unsigned b = first; b < n_bursts; b += total_waves) {
		// Re-materialise the lane id per burst: keeps the compiler from hoisting every lane-derived
		// address/predicate of every phase out of this loop (which cost ~70 spilled VGPRs).
		int lane_opaque = lane_outer;
		const int lane = lane_opaque;
		const trxhip_burst_params prm = params[b];
		const int type = uni(prm.type);
		const int tsc = uni(prm.tsc);
		int max_toa = uni(prm.max_toa);

		int rc = 0;
		float toa = 0.0f, ci = 0.0f, energy = 0.0f, rssi = 0.0f;
		c32 amp = make_float2(0.0f, 0.0f);
		int out_tsc = 0, clip = 0, idle = 1, nbits = 0;
		float *so = soft ? soft + (size_t)b * soft_stride : nullptr;

		// ---- phase 0: HBM -> fp32 LDS (convert_short_float fused into the load), clip scan
		float amax = 0.0f;
		if (NLD > 0) {
#pragma unroll
			for (int r = 0; r < NLD; r++) {
				const int i = r * WAVE + lane;
				if (i < L) {
					c32 v;
					if (CF32) v = pre_c[r];
					else v = make_float2((float)(int16_t)(pre_i[r] & 0xffffu), (float)(int16_t)(pre_i[r] >> 16));
					w.xs[TRX_PAD + i] = v;
					amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
				}
			}
			if (b + total_waves < n_bursts)
				prefetch(b + total_waves);
		} else if (CF32) {
			const c32 *src = reinterpret_cast<const c32 *>(iq_) + (size_t)b * L;
			for (int i = lane; i < L; i += WAVE) {
				const c32 v = src[i];
				w.xs[TRX_PAD + i] = v;
				amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
			}
		} else {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(iq_) + (size_t)b * L;
			for (int i = lane; i < L; i += WAVE) {
				const uint32_t u = src[i];
				const c32 v = make_float2((float)(int16_t)(u & 0xffffu), (float)(int16_t)(u >> 16));
				w.xs[TRX_PAD + i] = v;
				amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
			}
		}

		if (type != TRXHIP_OFF) {                                   // Transceiver.cpp:704-707
			amax = wave_max(amax);                                  // maxAmplitude(), :1711-1722
			clip = amax > TRX_CLIP_THRESH;
			wave_sync();

			// ---- energyDetect(burst, 20*sps) (:1573-1585): stride 4 regardless of sps; tree-summed
			{
				int win = 20 * SPS;
				if (win > L) win = L;
				float e = 0.0f;
				for (int i = lane; i < win; i += WAVE)
					e += norm2(w.xs[TRX_PAD + 4 * i]);
				energy = wave_sum(e) / (float)win;
				const float avg = sqrtf(energy);                    // Transceiver.cpp:741 (one path)
				rssi = (float)(20.0 * log10((double)full_scale / (double)avg));   // :751
			}

			if (type != TRXHIP_IDLE) {                              // Transceiver.cpp:754-755
				// ---- detectAnyBurst (:1926-1957): up to 3 candidate windows, first hit wins
				int ncand = 0;
				rc = 0;
				if (max_toa > TRXHIP_MAX_TOA) {
					rc = -TRXHIP_SIGERR_UNSUPPORTED;
				} else if (type == TRXHIP_TSC || type == TRXHIP_EDGE) {
					if (tsc > 7) rc = -TRXHIP_SIGERR_UNSUPPORTED;           // :1893, :1912
					else ncand = (type == TRXHIP_EDGE) ? 2 : 1;                // EDGE falls through to TSC (:1933-1941)
				} else if (type == TRXHIP_RACH || type == TRXHIP_EXT_RACH) {
					ncand = (type == TRXHIP_EXT_RACH) ? 3 : 1;                 // :1791
				}

				const c32 *sig;
				int sig_len;
				if (SPS == 4) {
					sig = w.dec;
					sig_len = 156;
				} else {
					sig = w.xs + TRX_PAD;
					sig_len = L;
				}

				int dec_lo = 1 << 30, dec_hi = 0;                 // decimated range already computed
				int det_type = 0;
				for (int c = 0; c < ncand; c++) {
					SeqWin cw;
					if (type == TRXHIP_RACH || type == TRXHIP_EXT_RACH)
						cw = { TRX_SEQ_RACH0 + c, 48, 8, 8 + max_toa };            // :1788-1790
					else if (type == TRXHIP_EDGE && c == 0)
						cw = { TRX_SEQ_EDGE0 + tsc, 82, 6, 6 + max_toa };           // :1915-1918
					else
						cw = { TRX_SEQ_TSC0 + tsc, 82, 10, 6 + max_toa };           // :1896-1899
					const trx_seq *sq = &tab->seq[cw.seq];
					const int N = sq->n;
					const int start = cw.target - cw.head - 1;             // :1752
					const int len = cw.head + cw.tail;                     // :1753

					if (SPS == 4) {
						// downsampleBurst (:1587-1601) restricted to what correlate/computeCI read:
						// dec[i] = sum_k xs[4i-15+k] * g[k], i in [lo, hi)
						int lo = start - (N - 1); if (lo < 0) lo = 0;
						int hi = start + len;     if (hi > 156) hi = 156;
						if (lo < dec_lo || hi > dec_hi) {
							for (int i = lo + lane; i < hi; i += WAVE) {
								const c32 *xp = w.xs + TRX_PAD + 4 * i - 15;
								float yr = 0.0f, yi = 0.0f;
#pragma unroll
								for (int k = 0; k < 16; k++) {
									const c32 x = xp[k];
									const float g = tab->dec_taps[k];
									yr += x.x * g;
									yi += x.y * g;
								}
								w.dec[i] = make_float2(yr, yi);
							}
							dec_lo = lo; dec_hi = hi;
							wave_sync();
						}
					}

					float t; c32 a; float cc;
					const int hit = detect_burst(sig, sig_len, w.corr + TRX_CZ_PAD, sq, thresh, start, len, sincv, lane, &t, &a, &cc);
					wave_sync();
					if (hit) {
						rc = 1;
						toa = t - (float)cw.head;                          // :1768
						amp = a;
						ci = cc;
						const int s = cw.seq;
						if (s >= TRX_SEQ_RACH0 && s < TRX_SEQ_RACH0 + 3) { out_tsc = s - TRX_SEQ_RACH0; det_type = type; }
						else if (s >= TRX_SEQ_EDGE0) { out_tsc = tsc; det_type = TRXHIP_EDGE; }
						else { out_tsc = tsc; det_type = TRXHIP_TSC; }
						break;
					}
				}
				if (rc > 0) rc = det_type;                                  // :1953-1954
				else if (rc == 0 && ncand > 0 && clip) rc = -TRXHIP_SIGERR_CLIP;   // :1764
			}
		}

		// ---- demodAnyBurst -> demodGmskBurst (:2055-2072) ----
		if (rc > 0 && rc != TRXHIP_EDGE) {
			// demodCommon (:2030-2048): delayVector(burst, -toa*sps), scaleVector(1/amp)
			const float delay = -toa * (float)SPS;
			const int whole = (int)floorf(delay);
			const float frac = delay - (float)whole;
			const bool use_filt = (double)fabsf(frac) > 1e-2;              // :1056
			const int fidx = use_filt ? (int)floorf(frac * (float)TRX_DELAY_FILTS) : 0;   // :1057
			const float *hf = tab->delay_filt[uni(fidx)];
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
				const c32 *xp = w.xs + TRX_PAD + mc - 9;
				if (use_filt) {
					// fshift[m] = sum_k X(m - 9 + k) * h[k]  (convolve NO_DELAY, 20 real taps; :1060)
					// tap-outer / output-inner: each output still accumulates k = 0..19 in order, but only a
					// sliding window of R samples (+ R accumulators) is live instead of all R+19 inputs
					constexpr int D = 3;                                    // LDS read-ahead, in taps
					c32 xr[R + 19];
#pragma unroll
					for (int j = 0; j < R; j++)
						yv[j] = make_float2(0.0f, 0.0f);
#pragma unroll
					for (int j = 0; j < R - 1 + D; j++)
						xr[j] = xp[j];
#pragma unroll
					for (int k = 0; k < 20; k++) {
						const float h = hf[k];
						if (R - 1 + D + k < R + 19)
							xr[R - 1 + D + k] = xp[R - 1 + D + k];
#pragma unroll
						for (int j = 0; j < R; j++) {
							yv[j].x += xr[j + k].x * h;
							yv[j].y += xr[j + k].y * h;
						}
						__builtin_amdgcn_sched_barrier(0);              // keep the window short: no load hoisting
					}
				} else {
#pragma unroll
					for (int j = 0; j < R; j++)
						yv[j] = xp[9 + j];
				}
#pragma unroll
				for (int j = 0; j < R; j++) {
					const int m = m0 + j;
					const bool ok = (mc == m0) && (m >= 0) && (m < L);
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
						w.xs[TRX_PAD + n0 + j] = yv[j];
			}
			wave_sync();

			// downsampleBurst (4 SPS) + GMSKReverseRotate + real part + vectorSlicer
			const int nsoft = (SPS == 4) ? 156 : L;
			nbits = 148;
			idle = 0;
			if (so) {
				const int nwrite = slice ? nbits : nsoft;
				for (int i = lane; i < soft_stride; i += WAVE) {
					float sv = 0.0f;
					if (i < nwrite) {
						c32 d;
						if (SPS == 4) {
							const c32 *xp = w.xs + TRX_PAD + 4 * i - 15;
							float yr = 0.0f, yi = 0.0f;
#pragma unroll
							for (int k = 0; k < 16; k++) {
								const c32 x = xp[k];
								const float g = tab->dec_taps[k];
								yr += x.x * g;
								yi += x.y * g;
							}
							d = make_float2(yr, yi);
						} else {
							d = w.xs[TRX_PAD + i];
						}
						const trx_c32 r = tab->rrot1[i];
						sv = r.re * d.x - r.im * d.y;                       // real(rot * x)  (:2066-2068)
						if (slice) {                                        // vectorSlicer (:546-556)
							float o = 0.5f * (sv + 1.0f);           // exact in fp32, as the double product
							if (o > 1.0f) o = 1.0f;
							else if (o < 0.0f) o = 0.0f;
							sv = o;
						}
					}
					so[i] = sv;
				}
			}
			wave_sync();
			// the in-place delayed burst leaves sample L-1 (and nothing else) stale: harmless, the next
			// burst overwrites [0, L) completely
		} else {
			if (rc == TRXHIP_EDGE) {
				// 8-PSK demodulation is not built yet: report detection only
				nbits = 0;
				idle = 0;
			}
			if (so)
				for (int i = lane; i < soft_stride; i += WAVE)
					so[i] = 0.0f;
		}

		// ---- result record: 32 bytes, one dword per lane 0..7
		if (lane < 8) {
			uint32_t word;
			switch (lane) {
			case 0: word = (uint32_t)rc; break;
			case 1: word = __float_as_uint(rc > 0 ? toa : 0.0f); break;
			case 2: word = __float_as_uint(rc > 0 ? amp.x : 0.0f); break;
			case 3: word = __float_as_uint(rc > 0 ? amp.y : 0.0f); break;
			case 4: word = __float_as_uint(rc > 0 ? ci : 0.0f); break;
			case 5: word = __float_as_uint(energy); break;
			case 6: word = __float_as_uint(rssi); break;
			default:
				word = (uint32_t)(rc > 0 ? out_tsc : 0) | ((uint32_t)clip << 8) | ((uint32_t)idle << 16) |
				       ((uint32_t)(nbits / 4) << 24);
				break;
			}
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
	return TRX_SINCV_LDS * sizeof(float) + (size_t)waves_per_block * slice_c32 * sizeof(c32);
}

extern "C" int trx_launch_pull(const void *d_iq, int cf32, const trxhip_burst_params *d_params,
			       trxhip_burst_result *d_results, float *d_soft, const trx_tables *d_tab,
			       size_t n_bursts, int L, int sps, float thresh, float full_scale,
			       int soft_stride, int slice, int n_cu, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	static int wpb_env = -1;
	if (wpb_env < 0) {
		const char *e = getenv("TRXHIP_WPB");           // tuning knob: waves (bursts in flight) per workgroup
		wpb_env = (e && atoi(e) == 8) ? 8 : 4;
	}
	const int wpb = wpb_env;
	const size_t lds = trx_pull_lds_bytes(L, wpb);
	if (lds > 160 * 1024)
		return TRXHIP_EINVAL;
	int blocks_per_cu = (int)((160 * 1024) / lds);
	if (blocks_per_cu > 8) blocks_per_cu = 8;
	if (blocks_per_cu < 1) blocks_per_cu = 1;
	size_t need = (n_bursts + wpb - 1) / wpb;
	size_t grid = (size_t)n_cu * blocks_per_cu;
	if (grid > need) grid = need;

#define LAUNCH(SPS_, CF_, NLD_)                                                                                 \
	do {                                                                                                    \
		auto k = (wpb == 8) ? burst_pull_kernel<SPS_, CF_, NLD_, 8> : burst_pull_kernel<SPS_, CF_, NLD_, 4>; \
		if (lds > 64 * 1024 &&                                                                           \
		    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
			return TRXHIP_EIO;                                                                      \
		hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(wpb * WAVE), lds, stream, d_iq, d_params, d_results, \
				   d_soft, d_tab, (unsigned)n_bursts, L, thresh, full_scale, soft_stride, slice); \
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
