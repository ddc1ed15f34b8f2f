// trx_va.hip -- the Viterbi alternative of pullRadioVector (cfg->use_va) for gfx950:
// scaleVector(burst, 1/16383) + demodAnyBurst_va() (Transceiver52M/Transceiver.cpp:782-784, :620-645) over gr-gsm's
// receiver in Transceiver52M/grgsm_vitac/: channel impulse response from the training sequence at 4 samples per
// symbol (grgsm_vitac.cpp:147-232, :244-272), matched filter (:168-181), 16-state MLSE (viterbi_detector.cc:62-392).
//
// Mapping: ONE WAVEFRONT PER BURST, four waves per workgroup, everything in that wave's LDS slice.
//   * the 59 training-sequence correlations: one lag per lane; |.|^2 the way libstdc++/glibc evaluate
//     std::pow(abs(c), 2): hypot in double, rounded to float, squared in double, rounded to float
//   * the sliding 20-sample energy window and its first maximum: the reference's serial float recurrence, kept
//     serial (59 steps on wave-uniform LDS reads) -- it decides the burst position
//   * matched filter: one output symbol per lane and round, 20 complex taps in order
//   * add-compare-select: lane = state (16 lanes), predecessors s>>1 and (s>>1)+8 fetched with a 16-wide shuffle,
//     the 32 hand-written ACS statements of the reference reduced to their sign/increment pattern; of every
//     path-metric difference only (d > 0, d != 0) matter for the +-127 output, so a step's table row is one ballot
//     word; the traceback is the reference's serial walk on the scalar unit
// Operand order follows the reference statement by statement (-ffp-contract=off): the +-127 outputs are bit-exact.
#include "trx_device.h"

#define VA_WPB 4
#define VA_SLICE_BYTES(xs_len) ((size_t)((xs_len) + 64 + VA_FL + VA_NB + 32 + 8) * sizeof(c32) + (size_t)(64 + 192) * sizeof(float))
#define VA_OSR 4
#define VA_CIR 5
#define VA_FL (VA_CIR * VA_OSR)
#define VA_NB 148
#define VA_AB 88

// 3GPP TS 45.002 training sequence bits (constants.h:91-95, :131-141 in the reference's grgsm_vitac/)
__constant__ char va_tsc_str[8][27] = {
	"00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110", "01000111101101000100011110",
	"00011010111001000001101011", "01001110101100000100111010", "10100111110110001010011111", "11101111000100101110111100",
};
__constant__ char va_acc_str[42] = "01001011011111111001100110101010001111000";

__device__ __forceinline__ c32 va_cmul(c32 a, c32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ void __launch_bounds__(VA_WPB * WAVE)
va_demod_kernel(const c32 *__restrict__ iq, const trxhip_burst_params *__restrict__ params,
		const trxhip_burst_result *__restrict__ detected, float *__restrict__ soft,
		int32_t *__restrict__ starts, unsigned n_bursts, int L, float scale, int soft_stride, int slice)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = uni((int)(threadIdx.x >> 6));
	const int xs_len = (L + 1) & ~1;
	// per-wave slice: xs[L] | corr[64] | cir[20] | filt[148] | seq[32] | rhh[8] : c32;  power[64] : float | tr[192] : u32
	const size_t slice_bytes = VA_SLICE_BYTES(xs_len);
	char *base = smem + (size_t)wave * slice_bytes;
	c32 *xs = reinterpret_cast<c32 *>(base);
	c32 *corr = xs + xs_len;
	c32 *cir = corr + 64;
	c32 *filt = cir + VA_FL;
	c32 *seq = filt + VA_NB;
	c32 *rhh = seq + 32;
	float *power = reinterpret_cast<float *>(rhh + 8);
	unsigned *tr = reinterpret_cast<unsigned *>(power + 64);   // decision words, one per step (entries >= 148 unused)

	const unsigned b = blockIdx.x * VA_WPB + wave;
	if (b >= n_bursts)
		return;
	const unsigned prm = reinterpret_cast<const uint32_t *>(params)[2 * (size_t)b];
	int type = prm & 0xff;
	const int tsc = (prm >> 8) & 0xff, max_toa = prm >> 16;
	float *so = soft + (size_t)b * soft_stride;
	bool skip = false;
	if (detected) {                                                // chained behind detection: rc is the CorrType (Transceiver.cpp:784)
		const int rc = uni(detected[b].rc);
		skip = rc <= 0;
		type = rc;
	}
	if (tsc > 7 || skip) {                                                 // train_seq has 8 entries (+ dummy): reject
		for (int i = lane; i < soft_stride; i += WAVE) so[i] = 0.0f;
		if (starts && lane == 0) starts[b] = -1;
		return;
	}
	const bool nb = (type == TRXHIP_TSC);                          // Transceiver.cpp:629: TSC, else the access branch
	const int nbits = nb ? VA_NB : VA_AB;

	// ---- scaleVector (sigProcLib.cpp:1198-1205): x * (scale, 0) with Complex.h:74's operand order
	const c32 *src = iq + (size_t)b * L;
	for (int i = lane; i < L; i += WAVE)
		xs[i] = va_cmul(src[i], make_float2(scale, 0.0f));

	// ---- training sequence, gmsk_mapper() + conj (grgsm_vitac.cpp:57-79, :122-145): a walk over {1, j, -1, -j}
	const int tlen = nb ? 26 : 41, tseqlen = tlen - 10;
	if (lane == 0) {
		const char *bits = nb ? va_tsc_str[tsc] : va_acc_str;
		int q = nb ? ((bits[0] == '0') ? 0 : 2) : 3;               // start point 1 / -1 (normal), -j (access)
		int prev = 2 * (bits[0] - '0') - 1;
		for (int i = 0; i < tlen; i++) {
			if (i > 0) {
				const int cur = 2 * (bits[i] - '0') - 1;
				q = (q + ((cur * prev > 0) ? 1 : 3)) & 3;          // times j or -j
				prev = cur;
			}
			const int qc = (4 - q) & 3;                            // conjugate
			if (i >= 5 && i < 5 + 32)
				seq[i - 5] = make_float2(qc == 0 ? 1.0f : qc == 2 ? -1.0f : 0.0f, qc == 1 ? 1.0f : qc == 3 ? -1.0f : 0.0f);
		}
	}
	wave_sync();

	// ---- get_chan_imp_resp (grgsm_vitac.cpp:183-232)
	const int center = nb ? (3 + 58 + 5) : (8 + 5);
	const int start_pos = (center - 5) * VA_OSR + 1, stop_pos = (center + 5 + VA_CIR) * VA_OSR;   // max_delay = 0 (:631)
	const int nw = stop_pos - start_pos;                           // 59
	if (lane < nw) {
		float rr = 0.0f, ri = 0.0f;
		for (int ii = 0; ii < tseqlen; ii++) {                     // correlate_sequence :147-155
			const int j = start_pos + lane + ii * VA_OSR;
			const c32 xv = (j < L) ? xs[j] : make_float2(0.0f, 0.0f);
			const c32 t = va_cmul(seq[ii], xv);
			rr += t.x;
			ri += t.y;
		}
		const float fl = (float)tseqlen;
		const c32 c = make_float2(rr / fl, -ri / fl);              // conj(result) / (length + 0j)
		corr[lane] = c;
		const float h = (float)sqrt((double)c.x * (double)c.x + (double)c.y * (double)c.y);   // abs(): hypotf
		power[lane] = (float)((double)h * (double)h);              // std::pow(float, int)
	}
	wave_sync();
	int best = 0;
	{
		float ws = 0.0f;
		for (int i = 0; i < VA_FL; i++)
			ws += power[i];
		float beste = ws;
		for (int i = VA_FL; i < nw; i++) {
			ws += power[i] - power[i - VA_FL];
			if (beste < ws) { beste = ws; best = i - (VA_FL - 1); }  // std::max_element: first largest
		}
		best = uni(best);
	}
	if (lane < VA_FL)
		cir[lane] = corr[best + lane];
	int start = start_pos + best - center * VA_OSR;
	if (start < 0) start = 0;                                      // Transceiver.cpp:631, :635
	wave_sync();

	// ---- detect_burst_generic (grgsm_vitac.cpp:82-108): rhh = conj(autocorrelation at multiples of 4), mafi
	if (lane < VA_CIR) {
		const int k = lane * VA_OSR;
		float ar = 0.0f, ai = 0.0f;
		for (int i = k; i < VA_FL; i++) {
			const c32 t = va_cmul(cir[i], make_float2(cir[i - k].x, -cir[i - k].y));
			ar += t.x;
			ai += t.y;
		}
		rhh[lane] = make_float2(ar, -ai);
	}
	for (int m = lane; m < nbits; m += WAVE) {
		float ar = 0.0f, ai = 0.0f;
		const int a = m * VA_OSR;
		for (int ii = 0; ii < VA_FL; ii++) {
			if (a + ii >= nbits * VA_OSR)
				break;
			const int j = start + a + ii;
			const c32 xv = (j < L) ? xs[j] : make_float2(0.0f, 0.0f);
			const c32 t = va_cmul(xv, cir[ii]);
			ar += t.x;
			ai += t.y;
		}
		filt[m] = make_float2(ar, ai);
	}
	wave_sync();

	// ---- viterbi_detector (viterbi_detector.cc:62-392): lane & 15 = state
	const int s = lane & 15, p = s >> 1, odd = s & 1;
	float inc[8];
#pragma unroll
	for (int m = 0; m < 8; m++) {
		float v = (m & 1) ? rhh[1].y : -rhh[1].y;
		v = (m & 2) ? v + rhh[2].x : v - rhh[2].x;
		v = (m & 4) ? v + rhh[3].y : v - rhh[3].y;
		inc[m] = v + rhh[4].x;
	}
	float ia = 0.0f, ib = 0.0f, ra = 0.0f, rb = 0.0f;              // this state's reference levels
#pragma unroll
	for (int m = 0; m < 8; m++) {
		ia = (m == (p ^ 2)) ? inc[m] : ia;                         // imaginary steps: inc[{2,3,0,1,6,7,4,5}[p]]
		ib = (m == (p ^ 5)) ? inc[m] : ib;                         //                  inc[{5,4,7,6,1,0,3,2}[p]]
		ra = (m == 7 - p) ? inc[m] : ra;                           // real steps:      inc[7 - p], inc[p]
		rb = (m == p) ? inc[m] : rb;
	}
	const unsigned start_state = nb ? 3u : (unsigned)max_toa;      // Transceiver.cpp:633: rach_max_toa as start state
	float pm = (-10e30);
	if ((unsigned)s == start_state)
		pm = 0.0f;
	// Only two bits of every path-metric difference survive into the +-127 output: d > 0 (the decision) and d != 0
	// (an output of +-0 is "not > 0").  Per step the 16 states' bits are one ballot word: tr[k] = nz << 16 | pos.
	for (int k = 0; k < nbits; k++) {
		const bool imag = !(k & 1);
		const c32 f = filt[k];
		const float sym = imag ? f.y : f.x;
		const float o1 = __shfl(pm, p, 16), o2 = __shfl(pm, p + 8, 16);
		// even state, imaginary step: o1 + sym - ia, o2 + sym + ib; odd: o1 - sym + ia, o2 - sym - ib
		// even state, real step:      o1 - sym - ra, o2 - sym + rb; odd: o1 + sym + ra, o2 + sym - rb
		const bool plus = imag ? !odd : odd;
		const float ss = plus ? sym : -sym;
		const float l1 = imag ? ia : ra, l2 = imag ? ib : rb;
		const float c1 = (o1 + ss) + (odd ? l1 : -l1);
		const float c2 = (o2 + ss) + (odd ? -l2 : l2);
		const float d = c2 - c1;
		pm = (d < 0) ? c1 : c2;
		const unsigned pos = (unsigned)__ballot(d > 0) & 0xffffu, nz = (unsigned)__ballot(d != 0) & 0xffffu;
		if (lane == 0)
			tr[k] = (nz << 16) | pos;
	}
	wave_sync();
	unsigned long long ones[3] = { 0ull, 0ull, 0ull };             // bit k of word k >> 6: output k is > 0
	{
		// best of the stop states {4, 12}; traceback with differential decoding (viterbi_detector.cc:340-392).
		// out[k] = +-d with the sign flipped when decision != out_bit, so out[k] > 0 <=> out_bit && d != 0.
		const float m4 = __shfl(pm, 4, 16), m12 = __shfl(pm, 12, 16);
		unsigned state = (unsigned)uni((m12 > m4) ? 12 : 4);
		unsigned out_bit = 0u, real_imag = (nbits & 1) ? 1u : 0u;  // type of the last step processed
		// the words are read back once, lane l holding steps l, l + 64, l + 128; the serial walk is scalar
		const unsigned wv[3] = { tr[lane], tr[lane + 64], (lane + 128 < VA_NB) ? tr[lane + 128] : 0u };
#pragma unroll
		for (int blk = 2; blk >= 0; blk--) {
			unsigned long long acc = 0ull;
			const int hi = (nbits - 1 < blk * 64 + 63) ? nbits - 1 - blk * 64 : 63;
			for (int kk = hi; kk >= 0; kk--) {
				const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wv[blk], kk);
				const unsigned decision = (w >> state) & 1u, nonzero = (w >> (16 + state)) & 1u;
				acc |= (unsigned long long)(out_bit & nonzero) << kk;
				const unsigned parity = ((state >> 1) ^ state) & 1u;
				out_bit = out_bit ^ real_imag ^ parity;
				state = (state >> 1) + (decision << 3);
				real_imag ^= 1u;
			}
			ones[blk] = acc;
		}
	}

	// ---- "pre flip" (:107), "* -1" (Transceiver.cpp:638), zeros behind the burst (:640-641); optional vectorSlicer
	for (int i = lane; i < soft_stride; i += WAVE) {
		float v = 0.0f;
		if (i < nbits) {
			const unsigned long long m = (i < 64) ? ones[0] : (i < 128) ? ones[1] : ones[2];
			v = ((m >> (i & 63)) & 1ull) ? 127.0f : -127.0f;
		}
		if (slice & 1)
			v = (i < 148) ? __builtin_amdgcn_fmed3f(0.5f * (v + 1.0f), 0.0f, 1.0f) : 0.0f;
		so[i] = v;
	}
	if (starts && lane == 0)
		starts[b] = start;
}

extern "C" size_t trx_va_lds_bytes(int L)
{
	const int xs_len = (L + 1) & ~1;
	return VA_WPB * VA_SLICE_BYTES(xs_len);
}

extern "C" int trx_launch_va_demod(const float *d_iq, const trxhip_burst_params *d_params,
				   const trxhip_burst_result *d_detected, float *d_soft, int32_t *d_starts,
				   size_t n_bursts, int L, float scale, int soft_stride, int flags, hipStream_t stream)
{
	if (n_bursts == 0)
		return 0;
	const size_t lds = trx_va_lds_bytes(L);
	if (lds > 160 * 1024)
		return TRXHIP_EINVAL;
	if (hipFuncSetAttribute((const void *)va_demod_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
		return TRXHIP_EIO;
	const size_t grid = (n_bursts + VA_WPB - 1) / VA_WPB;
	hipLaunchKernelGGL(va_demod_kernel, dim3((unsigned)grid), dim3(VA_WPB * WAVE), lds, stream,
			   reinterpret_cast<const c32 *>(d_iq), d_params, d_detected, d_soft, d_starts, (unsigned)n_bursts, L, scale, soft_stride,
			   flags);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
