// trx_va.hip -- the Viterbi alternative of pullRadioVector (cfg->use_va) for gfx950:
// scaleVector(burst, 1/16383) + demodAnyBurst_va() (Transceiver52M/Transceiver.cpp:782-784, :620-645) over gr-gsm's
// receiver in Transceiver52M/grgsm_vitac/: channel impulse response from the training sequence at 4 samples per
// symbol (grgsm_vitac.cpp:147-232, :244-272), matched filter (:168-181), 16-state MLSE (viterbi_detector.cc:62-392).
//
// Mapping: ONE WAVEFRONT PER FOUR BURSTS.  The front end (channel estimate, matched filter) runs burst after burst with
// all 64 lanes; the 16-state trellis and its traceback then run for the four bursts AT ONCE, one burst per DPP row of 16
// lanes -- the add-compare-select butterfly only ever talks to lanes of its own row, and the traceback, the reference's
// serial walk, is carried by the vector lanes of each row instead of the scalar unit (a scalar instruction costs a wave
// the same issue slot as a vector one, and the scalar walk could serve one burst only).
//   * the 59 training-sequence correlations: one lag per lane; |.|^2 the way libstdc++/glibc evaluate
//     std::pow(abs(c), 2): hypot in double, rounded to float, squared in double, rounded to float
//   * the sliding 20-sample energy window and its first maximum: the reference's serial float recurrence, kept
//     serial (59 steps on wave-uniform LDS reads) -- it decides the burst position
//   * matched filter: one output symbol per lane and round, 20 complex taps in order
//   * add-compare-select as a butterfly on 16 lanes: the lane holding old state S produces new state rotl4(S) from
//     its own metric and that of the lane holding S ^ 8, fetched with one DPP move (the state-to-lane map rotates
//     with period 4); the 32 hand-written ACS statements of the reference reduced to their sign/increment pattern; of every
//     path-metric difference only (d > 0, d != 0) matter for the +-127 output, so a step's table row is one ballot
//     word; the traceback is the reference's serial walk on the scalar unit
// Operand order follows the reference statement by statement (-ffp-contract=off): the +-127 outputs are bit-exact.
#include "trx_device.h"

#define VA_WPB 2                       // waves per workgroup (11.4 KB of LDS per wave: 14 waves per CU)
#define VA_BPW 4                       // bursts per wave: one per DPP row
#define VA_OSR 4
#define VA_CIR 5
#define VA_FL (VA_CIR * VA_OSR)
#define VA_NB 148
#define VA_AB 88
#define VA_FSTRIDE 152                 // floats per burst of trellis input symbols (148 + pad): the imaginary part of matched-filter
                                       // output k for even k, the real part for odd k -- all the trellis reads (one ds_read_b128 per 4 steps)
// The scaled burst is kept in a POLYPHASE layout, xs[ph * XA + i / 4] = x[i] with ph = i % 4: the matched filter walks
// the burst in steps of 4 samples per lane (one output symbol per lane), so consecutive lanes read consecutive words of
// one phase array (the linear layout was an 8-way bank conflict: 32-byte lane stride), and the training-sequence
// correlation (consecutive samples per lane) sees the four arrays 16 banks apart (XA = 8 mod 32).  Everything from x[L]
// to the end of the arrays is zero, which replaces the reference's "j < L" checks, and the arrays reach the last sample
// any stage can touch (start + 4 * 147 + 19 + 4 < 640) whatever L is.
#define VA_XA(L) (((((L) > 640 ? (L) : 640) + 3) / 4 + 31) / 32 * 32 + 8)
// per-wave LDS slice, every region 16-byte aligned:
//   scratch of the burst in the front end: xs[4][XA] | corr[64] | cir[20] | seq[32] : c32;  power[64] : float
//   kept for the trellis, per burst:       sym[4][152] : float | rhh[4][8] : c32;  meta[4] : int4 {nbits, start state, start, -}
//   decision words of the four trellises:  words[148] : uint4 {pos lo, pos hi, nz lo, nz hi} -- over xs[], which the
//                                          front end no longer needs by then
#define VA_SLICE_BYTES(L) ((size_t)(4 * VA_XA(L) + 64 + VA_FL + 32) * sizeof(c32) + 64 * sizeof(float) +                  \
			   (size_t)VA_BPW * (VA_FSTRIDE * sizeof(float) + 8 * sizeof(c32)) + VA_BPW * 16)

// The training sequences after gmsk_mapper() and conj() (grgsm_vitac.cpp:57-79, :122-145) are walks over
// {1, j, -1, -j}: out[i] = (+-j) * out[i-1] from the start point 1 / -1 (normal burst, first bit 0 / 1) or -j (access),
// then conjugated.  Stored as 2-bit quarter-turn codes (0: 1, 1: j, 2: -1, 3: -j) of the elements the channel
// estimate uses, i = 5 .. 20 of the 26 TSC bits and i = 5 .. 35 of the 41 access bits (TRAIN_BEGINNING = 5), element
// k at bits 2k, 2k+1 -- computed from the 3GPP TS 45.002 bit strings by the same walk (tests compare with the oracle,
// which maps the bits at run time).
#define VA_TSC_CODES0 0x131319b9ull
#define VA_TSC_CODES1 0x9311b9b9ull
#define VA_TSC_CODES2 0x1913b3b3ull
#define VA_TSC_CODES3 0x191933b1ull
#define VA_TSC_CODES4 0xbb191193ull
#define VA_TSC_CODES5 0x991b3391ull
#define VA_TSC_CODES6 0x139bb9b1ull
#define VA_TSC_CODES7 0x91933b31ull
#define VA_ACC_CODES 0x464e4ccccc6c644ull

// correlate_sequence() (grgsm_vitac.cpp:147-155) for one lag: sum_ii seq[ii] * x[j0 + 4 ii], seq[ii] a quarter turn.
// std::complex's a * b with a in {1, j, -1, -j} is b with its parts swapped / negated exactly (the products with 0 only
// contribute +-0), so every term of the reference's sum is ONE v_pk_add_f32 whose op_sel / neg modifiers carry the
// quarter turn (unit_mac<>, trx_device.h): code 0: (x, y), 1: (-y, x), 2: (-x, -y), 3: (y, -x).  p = the lane's address of
// x[j0] inside its phase array: the taps are 4 samples = 1 entry apart.
template <unsigned long long CODES, int N>
__device__ __forceinline__ trx_v2f va_corr(const c32 *p)
{
	trx_v2f acc = { 0.0f, 0.0f };
#pragma unroll
	for (int k0 = 0; k0 < N; k0 += 8) {
		c32 x[8];
#pragma unroll
		for (int u = 0; u < 8; u++)
			if (k0 + u < N)
				x[u] = lds_c32(p + k0 + u);
#pragma unroll
		for (int u = 0; u < 8; u++)
			if (k0 + u < N) {
				const unsigned c = (unsigned)((CODES >> (2 * (k0 + u))) & 3ull);
				const trx_v2f xv = { x[u].x, x[u].y };
				if (c == 0) acc = unit_mac<false, false>(acc, xv);
				if (c == 1) acc = unit_mac<true, false>(acc, xv);
				if (c == 2) acc = unit_mac<false, true>(acc, xv);
				if (c == 3) acc = unit_mac<true, true>(acc, xv);
			}
		__builtin_amdgcn_sched_barrier(0);
	}
	return acc;
}

__global__ void __launch_bounds__(VA_WPB * WAVE)
va_demod_kernel(const c32 *__restrict__ iq, const trxhip_burst_params *__restrict__ params,
		const trxhip_burst_result *__restrict__ detected, float *__restrict__ soft,
		int32_t *__restrict__ starts, unsigned n_bursts, int L, float scale, int soft_stride, int slice)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & (WAVE - 1);
	const int wave = uni((int)(threadIdx.x >> 6));
	const int XA = uni(VA_XA(L));
	const size_t slice_bytes = VA_SLICE_BYTES(L);
	char *base = smem + (size_t)wave * slice_bytes;
	c32 *xs = reinterpret_cast<c32 *>(base);                       // polyphase: sample i at xs[(i & 3) * XA + (i >> 2)]
	c32 *corr = xs + 4 * XA;
	c32 *cir = corr + 64;
	c32 *seq = cir + VA_FL;
	float *power = reinterpret_cast<float *>(seq + 32);
	c32 *prod = seq;                                               // 64 c32 over seq[] + power[]: the autocorrelation products
	float *sym_all = power + 64;
	c32 *rhh_all = reinterpret_cast<c32 *>(sym_all + VA_BPW * VA_FSTRIDE);
	int4 *meta = reinterpret_cast<int4 *>(rhh_all + VA_BPW * 8);
	uint4 *words = reinterpret_cast<uint4 *>(xs);                  // 148 x 16 bytes over the head of xs[] (4 * XA >= 672 entries)

	const unsigned q0 = (blockIdx.x * VA_WPB + wave) * VA_BPW;     // first burst of this wave
	if (q0 >= n_bursts)
		return;

	// =================== front end, one burst at a time (all 64 lanes) ===================
	for (int qb = 0; qb < VA_BPW; qb++) {
		const unsigned b = q0 + qb;
		float *sym = sym_all + qb * VA_FSTRIDE;
		c32 *rhh = rhh_all + qb * 8;
		if (b >= n_bursts) {                                       // batch tail: an idle row
			if (lane == 0) meta[qb] = make_int4(0, 0, -1, 0);
			continue;
		}
		const unsigned prm = reinterpret_cast<const uint32_t *>(params)[2 * (size_t)b];
		int type = prm & 0xff;
		const int tsc = (prm >> 8) & 0xff, max_toa = prm >> 16;
		float *so = soft + (size_t)b * soft_stride;
		bool skip = false;
		if (detected) {                                            // chained behind detection: rc is the CorrType (Transceiver.cpp:784)
			const int rc = uni(detected[b].rc);
			skip = rc <= 0;
			type = rc;
		}
		if (tsc > 7 || skip) {                                     // train_seq has 8 entries (+ dummy): reject
			for (int i = lane; i < soft_stride; i += WAVE) so[i] = 0.0f;
			if (starts && lane == 0) starts[b] = -1;
			if (lane == 0) meta[qb] = make_int4(0, 0, -1, 0);
			continue;
		}
		const bool nb = (type == TRXHIP_TSC);                      // Transceiver.cpp:629: TSC, else the access branch
		const int nbits = nb ? VA_NB : VA_AB;

		// ---- scaleVector (sigProcLib.cpp:1198-1205): x * (scale, 0).  Complex.h:74 evaluates (x.r*s - x.i*0, x.r*0 + x.i*s);
		// the products with 0 are +-0 and only ever decide the sign of a zero result, which nothing downstream can see
		// (sums, comparisons, |.|^2): two multiplies per sample instead of four and two additions.
		const c32 *src = iq + (size_t)b * L;
		{
			// sample lane + 64 r: phase lane & 3, entry (lane >> 2) + 16 r -- one address per lane, immediate offsets
			c32 *xp = xs + (lane & 3) * XA + (lane >> 2);
			for (int i = lane, r = 0; i < L; i += WAVE, r++) {
				const c32 v = src[i];
				xp[16 * r] = make_float2(v.x * scale, v.y * scale);
			}
		}
		for (int i = L + lane; i < 4 * XA; i += WAVE)              // zero behind the burst (the reference's "j < L ? x[j] : 0")
			xs[(i & 3) * XA + (i >> 2)] = make_float2(0.0f, 0.0f);
		wave_sync();

		// ---- get_chan_imp_resp (grgsm_vitac.cpp:183-232)
		const int center = nb ? (3 + 58 + 5) : (8 + 5);
		const int start_pos = (center - 5) * VA_OSR + 1, stop_pos = (center + 5 + VA_CIR) * VA_OSR;   // max_delay = 0 (:631)
		const int nw = stop_pos - start_pos;                       // 59
		{
			const int j0 = start_pos + (lane < nw ? lane : 0);
			const c32 *p = xs + (j0 & 3) * XA + (j0 >> 2);
			trx_v2f r;
			switch (nb ? tsc : 8) {                                // wave-uniform: one specialised loop per training sequence
			case 0: r = va_corr<VA_TSC_CODES0, 16>(p); break;
			case 1: r = va_corr<VA_TSC_CODES1, 16>(p); break;
			case 2: r = va_corr<VA_TSC_CODES2, 16>(p); break;
			case 3: r = va_corr<VA_TSC_CODES3, 16>(p); break;
			case 4: r = va_corr<VA_TSC_CODES4, 16>(p); break;
			case 5: r = va_corr<VA_TSC_CODES5, 16>(p); break;
			case 6: r = va_corr<VA_TSC_CODES6, 16>(p); break;
			case 7: r = va_corr<VA_TSC_CODES7, 16>(p); break;
			default: r = va_corr<VA_ACC_CODES, 31>(p); break;
			}
			// conj(result) / (length + 0j), length = tseqlen = 26 - 10 / 41 - 10: a division by 16 is an exact scaling
			const c32 c = nb ? make_float2(r.x * 0.0625f, -r.y * 0.0625f) : make_float2(r.x / 31.0f, -r.y / 31.0f);
			if (lane < nw) {
				corr[lane] = c;
				const float h = (float)sqrt((double)c.x * (double)c.x + (double)c.y * (double)c.y);   // abs(): hypotf
				power[lane] = (float)((double)h * (double)h);      // std::pow(float, int)
			}
		}
		wave_sync();
		// sliding 20-sample window energy (:199-214): ws = p[0] + ... + p[19], then ws += p[i] - p[i-20].  With q[i] = p[i]
		// (i < 20) or p[i] - p[i-20], the window sums are the left-to-right prefix sums of q: a serial DPP scan along the
		// lanes reproduces the reference's additions one for one (lane 19 + j ends with window j's energy).
		int best;
		{
			const float pw = (lane < nw) ? power[lane] : 0.0f;
			const float pp = (lane >= VA_FL && lane < nw) ? power[lane - VA_FL] : 0.0f;
			const float q = (lane < VA_FL) ? pw : pw - pp;
			float acc = q;
#pragma unroll
			for (int i = 1; i < 59; i++)                           // nw = 59 for both burst types
				asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q));
			const bool inwin = (lane >= VA_FL - 1) && (lane < nw);
			const float e = inwin ? acc : -3.0e38f;
			const float m = wave_max(e);
			const unsigned long long hit = __ballot(inwin && e == m);  // std::max_element: the first largest
			best = hit ? (int)__ffsll((unsigned long long)hit) - 1 - (VA_FL - 1) : 0;
		}
		if (lane < VA_FL)
			cir[lane] = corr[best + lane];
		int start = start_pos + best - center * VA_OSR;
		if (start < 0) start = 0;                                  // Transceiver.cpp:631, :635
		// the matched filter stops at sample start + 4 * nbits ("if (a + ii >= nbits * OSR) break", :99-100): the same
		// condition for every output, i.e. the samples from there on do not exist -- zero them (adding +-0 changes no sum)
		if (lane < VA_FL + 4) {
			const int j = start + nbits * VA_OSR + lane;
			xs[(j & 3) * XA + (j >> 2)] = make_float2(0.0f, 0.0f);
		}
		wave_sync();

		// ---- detect_burst_generic (grgsm_vitac.cpp:82-108): rhh = conj(autocorrelation at multiples of 4), mafi.
		// rhh[k] = conj(sum_{i >= 4k} cir[i] * conj(cir[i - 4k])): the 60 products in parallel (segment k = 20 - 4k of them,
		// parked over seq[] / power[], both dead by now), then lane k adds its segment in order.
		{
#pragma unroll
			for (int k = 0; k < VA_CIR; k++)
				if (lane >= k * VA_OSR && lane < VA_FL) {
					const c32 a = cir[lane], bb = cir[lane - k * VA_OSR];
					prod[(20 * k - 2 * k * (k - 1)) + lane - k * VA_OSR] = cmul(a, make_float2(bb.x, -bb.y));   // offsets 0, 20, 36, 48, 56
				}
			wave_sync();
			if (lane < VA_CIR) {
				const int seg = 20 * lane - 2 * lane * (lane - 1), len = VA_FL - VA_OSR * lane;
				float ar = 0.0f, ai = 0.0f;
#pragma unroll
				for (int i = 0; i < VA_FL; i++)
					if (i < len) {
						const c32 t = prod[seg + i];
						ar += t.x;
						ai += t.y;
					}
				rhh[lane] = make_float2(ar, -ai);
			}
			wave_sync();
		}
		{
			// mafi: filt[m] = sum_{ii < 20} x[start + 4m + ii] * cir[ii]: tap ii of every lane is phase (start + ii) & 3,
			// entry m + ((start + ii) >> 2) -- conflict-free, no range checks (zeros behind start + 4 nbits and behind L).
			// Taps in two halves of ten (registers), the three symbol rounds inside: every output still adds ii = 0 .. 19
			// in order.  Only one part of each output feeds the trellis: imaginary for even symbols, real for odd ones.
			trx_v2f acc[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
			for (int half = 0; half < 2; half++) {
				c32 hc[VA_FL / 2];
#pragma unroll
				for (int u = 0; u < VA_FL / 2; u++)
					hc[u] = cir[half * (VA_FL / 2) + u];           // wave-uniform: broadcast reads
#pragma unroll
				for (int rnd = 0; rnd < 3; rnd++) {
					if (rnd * WAVE < nbits) {                      // wave-uniform
						const int m = lane + rnd * WAVE;
						const int mc = m < VA_NB ? m : VA_NB - 1;      // lanes past the last symbol recompute it (not stored)
						// sample start + 4 mc + ii: phase (start + ii) & 3; four per-lane bases, then immediate offsets
						const c32 *pb[4];
#pragma unroll
						for (int k = 0; k < 4; k++)
							pb[k] = xs + ((start + k) & 3) * XA + ((start + k) >> 2) + mc;
#pragma unroll
						for (int u = 0; u < VA_FL / 2; u++) {
							const int ii = half * (VA_FL / 2) + u;
							const c32 xv = lds_c32(pb[ii & 3] + (ii >> 2));
							const c32 t = cmul(xv, hc[u]);
							acc[rnd] = acc[rnd] + (trx_v2f){ t.x, t.y };
						}
					}
				}
			}
#pragma unroll
			for (int rnd = 0; rnd < 3; rnd++) {
				const int m = lane + rnd * WAVE;
				if (m < nbits)
					sym[m] = (m & 1) ? acc[rnd].x : acc[rnd].y;
			}
		}
		if (lane == 0)                                             // Transceiver.cpp:633: rach_max_toa as the start state
			meta[qb] = make_int4(nbits, nb ? 3 : max_toa, start, 0);
		wave_sync();                                               // xs / corr / cir / seq are the next burst's scratch
	}
	wave_sync();

	// =================== viterbi_detector (viterbi_detector.cc:62-392) for the four bursts, row = burst ===================
	// Add-compare-select as a DPP butterfly.  New state n comes from old states p = n >> 1 and p + 8, i.e. the pair
	// (S, S ^ 8) feeds the pair rotl4(S), rotl4(S ^ 8).  So a lane holding old state S computes new state rotl4(S) from its
	// own metric and the metric of the lane holding S ^ 8: lane l of a row holds state rotl4^k(l) at step k (identity again
	// every 4 steps) and its partner is lane l ^ (8 >> (k & 3)) of the same row -- one or two row-local DPP moves.
	const int row = lane >> 4, l4 = lane & 15;
	const int4 mt = meta[row];
	const int nbits_row = mt.x;
	const float *mysym = sym_all + row * VA_FSTRIDE;
	const c32 *rhh = rhh_all + row * 8;
	float inc[8];
	{
		const float r1 = rhh[1].y, r2 = rhh[2].x, r3 = rhh[3].y, r4 = rhh[4].x;
#pragma unroll
		for (int m = 0; m < 8; m++) {
			float v = (m & 1) ? r1 : -r1;
			v = (m & 2) ? v + r2 : v - r2;
			v = (m & 4) ? v + r3 : v - r3;
			inc[m] = v + r4;
		}
	}
	// per layout r = k & 3: the state this lane holds, and for the state n = rotl4(S) it produces (p = S & 7,
	// odd = S >> 3) the signed reference levels and the sign of the input symbol:
	//   imaginary step (r even): even n: o1 + sym - inc[p^2], o2 + sym + inc[p^5];  odd n: o1 - sym + inc[p^2], o2 - sym - inc[p^5]
	//   real step      (r odd):  even n: o1 - sym - inc[7-p], o2 - sym + inc[p];    odd n: o1 + sym + inc[7-p], o2 + sym - inc[p]
	float a1[4], a2[4];
	unsigned sflip[4];                                             // sign-bit mask applied to the symbol
	bool oddr[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int S = ((l4 << r) | (l4 >> (4 - r))) & 15;          // rotl4^r(lane)
		const int p = S & 7;
		const bool odd = (S >> 3) != 0;
		float l1 = 0.0f, l2 = 0.0f;
#pragma unroll
		for (int m = 0; m < 8; m++) {
			l1 = (m == ((r & 1) ? 7 - p : (p ^ 2))) ? inc[m] : l1;
			l2 = (m == ((r & 1) ? p : (p ^ 5))) ? inc[m] : l2;
		}
		a1[r] = odd ? l1 : -l1;
		a2[r] = odd ? -l2 : l2;
		const bool plus = (r & 1) ? odd : !odd;
		sflip[r] = plus ? 0u : 0x80000000u;
		oddr[r] = odd;
	}
	float pm = (-10e30);
	if (l4 == mt.y)                                                // start state (>= 16 selects none, as in the reference's quirk)
		pm = 0.0f;
	const int nmax = max(max(uni(meta[0].x), uni(meta[1].x)), max(uni(meta[2].x), uni(meta[3].x)));
	// Only two bits of every path-metric difference survive into the +-127 output: d > 0 (the decision) and d != 0
	// (an output of +-0 is "not > 0").  Per step the 64 lanes' bits are two ballot words (row r = bits 16r .. 16r + 15,
	// bit l = the new state rotl4^(k+1)(l)), parked in LDS for the traceback.
	float4 f4 = *reinterpret_cast<const float4 *>(mysym);          // symbols 0 .. 3 of this row's burst
	for (int k0 = 0; k0 < nmax; k0 += 4) {                         // 148 and 88 are multiples of 4
		const float4 fc = f4;
		f4 = *reinterpret_cast<const float4 *>(mysym + k0 + 4);    // next group in flight while this one runs (pad: 152 entries)
		const bool act = k0 < nbits_row;                           // this row's burst is still running (rows may differ in length)
		unsigned long long posw[4], nzw[4];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const float sym = (r == 0) ? fc.x : (r == 1) ? fc.y : (r == 2) ? fc.z : fc.w;
			const int pmi = __float_as_int(pm);
			int other;
			if (r == 0)      other = __builtin_amdgcn_update_dpp(pmi, pmi, 0x128, 0xf, 0xf, false);   // row_ror:8   (l ^ 8)
			else if (r == 1) {                                                                           // l ^ 4
				other = __builtin_amdgcn_update_dpp(pmi, pmi, 0x104, 0xf, 0x5, false);                   // row_shl:4 into banks 0, 2
				other = __builtin_amdgcn_update_dpp(other, pmi, 0x114, 0xf, 0xa, false);                 // row_shr:4 into banks 1, 3
			}
			else if (r == 2) other = __builtin_amdgcn_update_dpp(pmi, pmi, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1] (l ^ 2)
			else             other = __builtin_amdgcn_update_dpp(pmi, pmi, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2] (l ^ 1)
			const float po = __int_as_float(other);
			const float o1 = oddr[r] ? po : pm, o2 = oddr[r] ? pm : po;
			const float ss = __int_as_float(__float_as_int(sym) ^ (int)sflip[r]);
			const float c1 = (o1 + ss) + a1[r];
			const float c2 = (o2 + ss) + a2[r];
			const float d = c2 - c1;
			const float npm = (d < 0) ? c1 : c2;
			pm = act ? npm : pm;
			posw[r] = __ballot(d > 0);
			nzw[r] = __ballot(d != 0);
		}
		if (lane == 0) {                                           // one masked block per four steps
#pragma unroll
			for (int r = 0; r < 4; r++)
				words[k0 + r] = make_uint4((unsigned)posw[r], (unsigned)(posw[r] >> 32), (unsigned)nzw[r], (unsigned)(nzw[r] >> 32));
		}
	}
	wave_sync();

	// ---- best of the stop states {4, 12}; traceback with differential decoding (viterbi_detector.cc:340-392), every
	// lane of a row walking its row's path.  out[k] = +-d with the sign flipped when decision != out_bit, so
	// out[k] > 0 <=> out_bit && d != 0.  After a multiple of 4 steps lane l of a row holds state l again.
	const float m4 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & ~15) | 4) << 2, __float_as_int(pm)));
	const float m12 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & ~15) | 12) << 2, __float_as_int(pm)));
	// The walk tracks, instead of the state s_k, the LANE of the row that produced s_k's decision at step k,
	// l_k = rotr4^(k+1)(s_k) (new state n of step k sits at lane rotr4^(k+1)(n)): s_(k-1) = (s_k >> 1) + (decision << 3) is
	// rotr4(s_k) with bit 3 replaced by the decision, hence l_(k-1) = l_k with bit q = (3 - k) & 3 replaced by it -- no
	// rotation per step; bits 0 and 1 of s_k (its parity) sit at bits q and (q + 1) & 3 of l_k.  nbits = 0 mod 4: l = s at
	// the start.  The row's 16 decision / non-zero bits of step k are the u16 at bytes 2 row / 8 + 2 row of words[k].
	unsigned ln = (m12 > m4) ? 12u : 4u;
	unsigned out_bit = 0u;                                         // only bit 0 is meaningful (masked by `nonzero` where used)
	unsigned ones[5] = { 0u, 0u, 0u, 0u, 0u };                     // bit k & 31 of word k >> 5: output k is > 0
	const unsigned short *w16 = reinterpret_cast<const unsigned short *>(words) + row;
#pragma unroll
	for (int wq = 4; wq >= 0; wq--) {
		if (32 * wq >= nmax)
			continue;
		const int khi = (nmax - 1 < 32 * wq + 31) ? nmax - 1 - 32 * wq : 31;
		unsigned acc = 0u;
		unsigned pnext = w16[8 * (32 * wq + khi)], nnext = w16[8 * (32 * wq + khi) + 4];   // one step ahead
		for (int kk = khi; kk >= 0; kk--) {
			const int k = 32 * wq + kk;
			const unsigned pw = pnext, nw2 = nnext;
			const int kp = k > 0 ? k - 1 : 0;
			pnext = w16[8 * kp];
			nnext = w16[8 * kp + 4];
			if (k < nbits_row) {
				// type of step k: the last step processed is step nbits - 1 with real_imag = nbits & 1 = 0 (148 and 88 are
				// even) and the flag alternates, so real_imag(k) = (nbits - 1 - k) & 1 = (k + 1) & 1
				const unsigned real_imag = (unsigned)(k + 1) & 1u;
				const unsigned q = (unsigned)(3 - k) & 3u, q1 = (q + 1u) & 3u;
				const unsigned decision = (pw >> ln) & 1u, nonzero = (nw2 >> ln) & 1u;
				acc |= (out_bit & nonzero) << kk;
				out_bit = out_bit ^ real_imag ^ (ln >> q) ^ (ln >> q1);
				ln = (ln & ~(1u << q)) | (decision << q);
			}
		}
		ones[wq] = acc;
	}

	// ---- "pre flip" (:107), "* -1" (Transceiver.cpp:638), zeros behind the burst (:640-641); optional vectorSlicer.
	// Lane l of a row writes outputs l, l + 16, ... of its burst.
	const unsigned bme = q0 + row;
	if (bme < n_bursts && nbits_row > 0) {
		float *so = soft + (size_t)bme * soft_stride;
#pragma unroll
		for (int t = 0; t < 10; t++) {                             // outputs 0 .. 159; bit (16 t + l4) of the 160-bit string
			const int i = 16 * t + l4;
			float v = 0.0f;
			if (i < nbits_row)
				v = ((ones[t >> 1] >> (16 * (t & 1) + l4)) & 1u) ? 127.0f : -127.0f;
			if (slice & 1)
				v = (i < 148) ? __builtin_amdgcn_fmed3f(0.5f * (v + 1.0f), 0.0f, 1.0f) : 0.0f;
			if (i < soft_stride)
				so[i] = v;
		}
		for (int i = 160 + l4; i < soft_stride; i += 16)
			so[i] = (slice & 1) ? 0.0f : 0.0f;
		if (starts && l4 == 0)
			starts[bme] = mt.z;
	}
}

extern "C" size_t trx_va_lds_bytes(int L)
{
	return VA_WPB * VA_SLICE_BYTES(L);
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
	TRX_ARM_DYNAMIC_LDS(va_demod_kernel);
	const size_t per_block = VA_WPB * VA_BPW;
	const size_t grid = (n_bursts + per_block - 1) / per_block;
	hipLaunchKernelGGL(va_demod_kernel, dim3((unsigned)grid), dim3(VA_WPB * WAVE), lds, stream,
			   reinterpret_cast<const c32 *>(d_iq), d_params, d_detected, d_soft, d_starts, (unsigned)n_bursts, L, scale, soft_stride,
			   flags);
	return hipGetLastError() == hipSuccess ? 0 : TRXHIP_EIO;
}
