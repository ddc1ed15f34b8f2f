/*
 * trx_oracle.c -- CPU restatement (plain C) of osmo-trx's receive-side burst DSP.
 *
 * TEST INFRASTRUCTURE ONLY (see trx_oracle.h).  Written from the behaviour of the reference,
 * function by function, keeping the reference's operand order and float/double promotion points
 * (the C++ overloads the reference resolves to are noted where C would differ).
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile); never -ffast-math.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "trx_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* sigProcLib.cpp:55  static const float M_PI_F = (float)M_PI; */
static const float M_PI_F = (float)M_PI;
/* sigProcLib.cpp:49 */
#define CLIP_THRESH 30000.0f

static orc_tables T;
static int T_ready;

/* ------------------------------------------------------------------------------------------
 * Complex<float> helpers, Transceiver52M/Complex.h
 * ---------------------------------------------------------------------------------------- */
static inline orc_cf cf(float re, float im) { orc_cf z; z.re = re; z.im = im; return z; }
/* Complex.h:74  operator*(Complex): (r*a.r - i*a.i, r*a.i + i*a.r) */
static inline orc_cf cmul(orc_cf x, orc_cf a) { return cf(x.re * a.re - x.im * a.im, x.re * a.im + x.im * a.re); }
/* Complex.h:75  operator*(Real) */
static inline orc_cf cmulr(orc_cf x, float a) { return cf(x.re * a, x.im * a); }
/* Complex.h:113 norm2(): i*i + r*r */
static inline float cnorm2(orc_cf x) { return x.im * x.im + x.re * x.re; }
/* Complex.h:123 abs(): ::sqrt(norm2()) -- float overload */
static inline float cabs_(orc_cf x) { return sqrtf(cnorm2(x)); }
/* Complex.h:144-150 inv(): (r/n, -i/n) */
static inline orc_cf cinv(orc_cf x) { float n = cnorm2(x); return cf(x.re / n, -x.im / n); }
/* Complex.h:76  operator/(Complex) = operator*(a.inv()) */
static inline orc_cf cdiv(orc_cf x, orc_cf a) { return cmul(x, cinv(a)); }

/* ------------------------------------------------------------------------------------------
 * arch/common/convolve_base.c:28-85 -- generic C FIR, correlation form (no tap flip)
 * ---------------------------------------------------------------------------------------- */
int orc_convolve_real(const float *x, int x_len, const float *h, int h_len,
		      float *y, int y_len, int start, int len)
{
	(void)x_len; (void)y_len;
	memset(y, 0, (size_t)len * 2 * sizeof(float));   /* arch/x86/convolve.c:102 */
	for (int i = 0; i < len; i++) {
		const float *xp = &x[2 * (i - (h_len - 1) + start)];
		for (int k = 0; k < h_len; k++) {
			y[2 * i + 0] += xp[2 * k + 0] * h[2 * k];   /* mac_real :28-32 */
			y[2 * i + 1] += xp[2 * k + 1] * h[2 * k];
		}
	}
	return len;
}

int orc_convolve_complex(const float *x, int x_len, const float *h, int h_len,
			 float *y, int y_len, int start, int len)
{
	(void)x_len; (void)y_len;
	memset(y, 0, (size_t)len * 2 * sizeof(float));   /* arch/x86/convolve.c:142 */
	for (int i = 0; i < len; i++) {
		const float *xp = &x[2 * (i - (h_len - 1) + start)];
		for (int k = 0; k < h_len; k++) {
			/* mac_cmplx :35-39 */
			y[2 * i + 0] += xp[2 * k + 0] * h[2 * k + 0] - xp[2 * k + 1] * h[2 * k + 1];
			y[2 * i + 1] += xp[2 * k + 0] * h[2 * k + 1] + xp[2 * k + 1] * h[2 * k + 0];
		}
	}
	return len;
}

/* arch/common/convert_base.c:27-31 */
void orc_convert_short_float(float *out, const short *in, int len)
{
	for (int i = 0; i < len; i++)
		out[i] = in[i];
}

/* arch/common/convert_base.c:21-25 (truncating cast) */
void orc_convert_float_short(short *out, const float *in, float scale, int len)
{
	for (int i = 0; i < len; i++)
		out[i] = (short)(in[i] * scale);
}

/*
 * Optional arch seam (tests and bench.py's second CPU baseline only).  orc_set_arch() routes every FIR of the call graph --
 * table generation included -- through externally supplied kernels with the reference's signatures
 * (arch/common/convolve.h:6-14: convolve_real / convolve_complex): tests plug the reference's OWN objects, compiled
 * unmodified into oracle/_ref/libref_generic.so / libref_sse.so (oracle/Makefile), in here.  With the generic build every
 * result must equal the restated loops below bit for bit (tests/test_oracle.py): that pins the loops, the zero padding and
 * the call arguments of every site against the reference's kernels inside the full detect / demod call graph.  With the SSE
 * build the oracle becomes the reference's SSE path (north star: "match the reference CPU/SSE path").
 * NULL, NULL restores the restated loops.  Tables are rebuilt by the next call (they are convolution outputs themselves).
 */
static orc_conv_fn A_real, A_cmplx;

void orc_set_arch(orc_conv_fn conv_real, orc_conv_fn conv_complex)
{
	A_real = conv_real;
	A_cmplx = conv_complex;
	T_ready = 0;
}

/* what the reference's convolve() does in front of the kernel call (sigProcLib.cpp:309-358): a copy of x with `head` zero
 * samples in front and `tail` behind when the span reaches outside the vector, taps in a 16-byte aligned buffer
 * (convolve_h_alloc); the kernel is then called with start moved by `head`.  `aligned` = h->isAligned() of that call site:
 * false selects base_convolve_* in the reference (:374-383), i.e. the generic loops, whatever kernels are plugged. */
static int conv_arch(const orc_cf *x, int n, const float *h2, int H, int h_real, orc_cf *y, int start, int len)
{
	int head = (start < H - 1) ? (H - 1 - start) : 0;
	int tail = (start + len > n) ? (start + len - n) : 0;
	size_t tot = (size_t)head + (size_t)n + (size_t)tail;
	float *xp = NULL, *hp = NULL, *yp = NULL;
	int rc = -1;
	if (posix_memalign((void **)&xp, 16, (tot + 4) * 2 * sizeof(float)) ||
	    posix_memalign((void **)&hp, 16, ((size_t)H + 4) * 2 * sizeof(float)) ||
	    posix_memalign((void **)&yp, 16, ((size_t)len + 4) * 2 * sizeof(float)))
		goto out;
	memset(xp, 0, (tot + 4) * 2 * sizeof(float));
	memcpy(xp + 2 * (size_t)head, x, (size_t)n * sizeof(orc_cf));
	memcpy(hp, h2, (size_t)H * 2 * sizeof(float));
	rc = (h_real ? A_real : A_cmplx)(xp, (int)tot, hp, H, yp, len, start + head, len);
	memcpy(y, yp, (size_t)len * sizeof(orc_cf));
out:
	free(xp); free(hp); free(yp);
	return rc;
}

/*
 * sigProcLib.cpp:297-398 convolve() reduced to what every span type computes:
 *   y[i] = sum_k X(i + start - (H-1) + k) * h[k],  X = x inside [0,n) and 0 outside
 * (START_ONLY reads zeroed head-room, NO_DELAY/CUSTOM zero-pad a copy :352-353).
 * h_real: taps are real (mac_real), else complex (mac_cmplx).  Sequential k order.
 * aligned: h->isAligned() at this call site (only matters when arch kernels are plugged, see above).
 */
static void conv_span(const orc_cf *x, int n, const orc_cf *h, int H, int h_real,
		      orc_cf *y, int start, int len, int aligned)
{
	if (aligned && (h_real ? A_real : A_cmplx) && len > 0) {
		conv_arch(x, n, (const float *)h, H, h_real, y, start, len);
		return;
	}
	for (int i = 0; i < len; i++) {
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < H; k++) {
			int j = i + start - (H - 1) + k;
			orc_cf xv = (j >= 0 && j < n) ? x[j] : cf(0.0f, 0.0f);
			if (h_real) {
				yr += xv.re * h[k].re;
				yi += xv.im * h[k].re;
			} else {
				yr += xv.re * h[k].re - xv.im * h[k].im;
				yi += xv.re * h[k].im + xv.im * h[k].re;
			}
		}
		y[i] = cf(yr, yi);
	}
}

static void conv_span_rtaps(const orc_cf *x, int n, const float *h, int H,
			    orc_cf *y, int start, int len, int aligned)
{
	if (aligned && A_real && len > 0 && H <= 64) {
		float h2[2 * 64];
		for (int k = 0; k < H; k++) { h2[2 * k] = h[k]; h2[2 * k + 1] = 0.0f; }   /* real taps live in complex storage */
		conv_arch(x, n, h2, H, 1, y, start, len);
		return;
	}
	for (int i = 0; i < len; i++) {
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < H; k++) {
			int j = i + start - (H - 1) + k;
			orc_cf xv = (j >= 0 && j < n) ? x[j] : cf(0.0f, 0.0f);
			yr += xv.re * h[k];
			yi += xv.im * h[k];
		}
		y[i] = cf(yr, yi);
	}
}

/* ------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------- */
/* GSM/GSMCommon.cpp:35-68 -- 3GPP TS 45.002 constants */
static const char *const TSC_BITS[8] = {
	"00100101110000100010010111", "00101101110111100010110111",
	"01000011101110100100001110", "01000111101101000100011110",
	"00011010111001000001101011", "01001110101100000100111010",
	"10100111110110001010011111", "11101111000100101110111100",
};
static const char *const EDGE_TSC_BITS[8] = {
	"111111001111111001111001001001111111111111001111111111001111111001111001001001",
	"111111001111001001111001001001111001001001001111111111001111001001111001001001",
	"111001111111111111001001001111001001001111001111111001111111111111001001001111",
	"111001111111111001001001001111001001111001111111111001111111111001001001001111",
	"111111111001001111001111001001001111111001111111111111111001001111001111001001",
	"111001111111001001001111001111001001111111111111111001111111001001001111001111",
	"001111001111111001001001001001111001001111111111001111001111111001001001001001",
	"001001001111001001001001111111111001111111001111001001001111001001001001111111",
};
static const char *const DUMMY_TSC_BITS = "01110001011100010111000101";
static const char *const RACH_BITS[3] = {
	"01001011011111111001100110101010001111000",
	"01010100111110001000011000101111001001101",
	"11101111001001110101011000001101101110111",
};
static const char *const SCH_BITS =
	"1011100101100010000001000000111100101101010001010111011000011011";

static int bits_from_str(const char *s, uint8_t *out)
{
	int n = 0;
	while (s[n]) { out[n] = (uint8_t)(s[n] == '1'); n++; }
	return n;
}

/* sigProcLib.cpp:981-988 */
static void generate_sinc_table(void)
{
	for (int i = 0; i < ORC_SINC_TABLESIZE; i++) {
		double x = (double)i / ORC_SINC_TABLESIZE * 8 * M_PI;
		double y = sin(x) / x;
		T.sinc_table[i] = isnan(y) ? 1.0 : y;
	}
	T.sinc_table[ORC_SINC_TABLESIZE] = 0.0f; /* static storage, never written (:52) */
}

/* sigProcLib.cpp:990-998.  C++ fabs(float) is the float overload; promoted to double by the
 * comparison / division against the double 8*M_PI; floorf() takes the double rounded to float. */
static float sinc_lut(float x)
{
	if ((double)fabsf(x) >= 8 * M_PI)
		return 0.0f;
	int index = (int)floorf((float)((double)fabsf(x) / (8 * M_PI) * ORC_SINC_TABLESIZE));
	return T.sinc_table[index];
}

/* sigProcLib.cpp:191-216 */
static void init_gmsk_rotation_tables(void)
{
	double phase = 0.0;
	for (int i = 0; i < 625; i++) {
		T.rot4[i]  = cf((float)cos(phase),  (float)sin(phase));
		T.rrot4[i] = cf((float)cos(-phase), (float)sin(-phase));
		phase += M_PI / 2.0 / 4.0;
	}
	phase = 0.0;
	for (int i = 0; i < 157; i++) {
		T.rot1[i]  = cf((float)cos(phase),  (float)sin(phase));
		T.rrot1[i] = cf((float)cos(-phase), (float)sin(-phase));
		phase += M_PI / 2.0;
	}
}

/* sigProcLib.cpp:463-543 (+ :405-461) */
static void generate_gsm_pulses(void)
{
	static const double c0_4[16] = {
		0.0, 4.46348606e-03, 2.84385729e-02, 1.03184855e-01, 2.56065552e-01, 4.76375085e-01,
		7.05961177e-01, 8.71291644e-01, 9.29453645e-01, 8.71291644e-01, 7.05961177e-01,
		4.76375085e-01, 2.56065552e-01, 1.03184855e-01, 2.84385729e-02, 4.46348606e-03 };
	static const double c1_4[8] = {
		0.0, 8.16373112e-03, 2.84385729e-02, 5.64158904e-02, 7.05463553e-02, 5.64158904e-02,
		2.84385729e-02, 8.16373112e-03 };
	static const double inv[5] = { 0.15884, -0.43176, 1.00000, -0.42608, 0.14882 };

	for (int i = 0; i < 16; i++) T.pulse4_c0[i] = (float)c0_4[i];
	for (int i = 0; i < 8; i++)  T.pulse4_c1[i] = (float)c1_4[i];
	for (int i = 0; i < 5; i++)  T.c0_inv[i] = (float)inv[i];

	/* 1 SPS: :519-533 */
	int len = 4, sps = 1;
	float center = (float)(len - 1.0) / 2.0;
	for (int i = 0; i < len; i++) {
		float arg = ((float)i - center) / (float)sps;
		T.pulse1_c0[i] = (float)(0.96 * exp(-1.1380 * arg * arg - 0.527 * arg * arg * arg * arg));
	}
	float energy = 0.0f;                    /* vectorNorm2 :178-186 (imag parts are 0) */
	for (int i = 0; i < len; i++)
		energy += 0.0f * 0.0f + T.pulse1_c0[i] * T.pulse1_c0[i];
	float avg = sqrtf(energy / sps);
	for (int i = 0; i < len; i++)
		T.pulse1_c0[i] /= avg;
}

/* sigProcLib.cpp:218-260 GMSKRotate, non-NEON branch */
static void gmsk_rotate(orc_cf *x, int n, int sps, int is_real)
{
	const orc_cf *rot = (sps == 1) ? T.rot1 : T.rot4;
	for (int i = 0; i < n; i++) {
		if (is_real)
			x[i] = cmulr(rot[i], x[i].re);    /* *rotPtr * xPtr->real() */
		else
			x[i] = cmul(rot[i], x[i]);        /* *rotPtr * *xPtr */
	}
}

/* sigProcLib.cpp:558-580 rotateBurst (emptyPulse): rotate +-1 symbols, 1-tap "filter" */
static int rotate_burst(const uint8_t *bits, int nbits, int guard, int sps, orc_cf *out)
{
	int burst_len = sps * (nbits + guard);
	orc_cf *rot = calloc((size_t)burst_len, sizeof(orc_cf));
	for (int i = 0; i < nbits; i++)
		rot[i * sps] = cf((float)(2.0 * (bits[i] & 0x01) - 1.0), 0.0f);
	gmsk_rotate(rot, burst_len, sps, 0);         /* `rotated` is not flagged real */
	orc_cf one = cf(1.0f, 0.0f);
	conv_span(rot, burst_len, &one, 1, 1, out, 0, burst_len, 0);  /* START_ONLY, h = empty pulse (not aligned) */
	free(rot);
	return burst_len;
}

/* sigProcLib.cpp:938-967 modulateBurstBasic */
static int modulate_burst_basic(const uint8_t *bits, int nbits, int guard, int sps, orc_cf *out)
{
	const float *pulse = (sps == 1) ? T.pulse1_c0 : T.pulse4_c0;
	int plen = (sps == 1) ? 4 : 16;
	int burst_len = sps * (nbits + guard);
	orc_cf *b = calloc((size_t)burst_len, sizeof(orc_cf));
	for (int i = 0; i < nbits; i++)
		b[i * sps] = cf((float)(2.0 * (bits[i] & 0x01) - 1.0), 0.0f);
	gmsk_rotate(b, burst_len, sps, 1);
	conv_span_rtaps(b, burst_len, pulse, plen, out, 0, burst_len, 1);   /* START_ONLY; c0 is aligned (:497) */
	free(b);
	return burst_len;
}

/* sigProcLib.cpp:595-670 modulateBurstLaurent (4 SPS, 625 samples) */
static int modulate_burst_laurent(const uint8_t *bits, int nbits, orc_cf *out)
{
	const int sps = 4, burst_len = 625;
	if (nbits > 156 || nbits < 2)
		return -1;
	orc_cf *c0 = calloc(burst_len, sizeof(orc_cf));
	orc_cf *c1 = calloc(burst_len, sizeof(orc_cf));
	orc_cf *c1s = calloc(burst_len, sizeof(orc_cf));
	int p = 0;
	c0[p] = cf((float)(2.0 * (0x00 & 0x01) - 1.0), 0.0f);           /* :618 */
	p += sps;
	for (int i = 0; i < nbits; i++) {                                /* :622-625 */
		c0[p] = cf((float)(2.0 * (bits[i] & 0x01) - 1.0), 0.0f);
		p += sps;
	}
	c0[p] = cf((float)(2.0 * (0x00 & 0x01) - 1.0), 0.0f);           /* :628 */
	gmsk_rotate(c0, burst_len, sps, 1);                              /* :631 */

	int q = sps * 2;                                                 /* :634-636 */
	float phase = (float)(2.0 * ((0x01 & 0x01) ^ (0x01 & 0x01)) - 1.0);   /* :639 */
	c1[q] = cmul(c0[q], cf(0.0f, phase));
	q += sps;
	for (int i = 2; i < nbits; i++) {                                /* :645-651 */
		phase = (float)(2.0 * ((bits[i - 1] & 0x01) ^ (bits[i - 2] & 0x01)) - 1.0);
		c1[q] = cmul(c0[q], cf(0.0f, phase));
		q += sps;
	}
	phase = (float)(2.0 * ((bits[nbits - 1] & 0x01) ^ (bits[nbits - 2] & 0x01)) - 1.0);  /* :654-656 */
	c1[q] = cmul(c0[q], cf(0.0f, phase));

	conv_span_rtaps(c0, burst_len, T.pulse4_c0, 16, out, 0, burst_len, 1);   /* :659 */
	conv_span_rtaps(c1, burst_len, T.pulse4_c1, 8, c1s, 0, burst_len, 1);    /* :660; c1 aligned (:443) */
	for (int i = 0; i < burst_len; i++) {                                  /* :663-666 */
		out[i].re += c1s[i].re;
		out[i].im += c1s[i].im;
	}
	free(c0); free(c1); free(c1s);
	return burst_len;
}

/* sigProcLib.cpp:970-979 */
int orc_modulate_burst(const uint8_t *bits, int nbits, int guard, int sps, int empty_pulse, orc_cf *out)
{
	if (!T_ready) orc_setup();
	if (empty_pulse)
		return rotate_burst(bits, nbits, guard, sps, out);
	else if (sps == 4)
		return modulate_burst_laurent(bits, nbits, out);
	else
		return modulate_burst_basic(bits, nbits, guard, sps, out);
}

/* sigProcLib.cpp:66-75 */
static const double PSK8[8][2] = {
	{ -0.70710678,  0.70710678 }, { 0.0, -1.0 }, { 0.0, 1.0 }, { 0.70710678, -0.70710678 },
	{ -1.0, 0.0 }, { -0.70710678, -0.70710678 }, { 0.70710678, 0.70710678 }, { 1.0, 0.0 },
};

/* sigProcLib.cpp:713-729 mapEdgeSymbols */
static int map_edge_symbols(const uint8_t *bits, int nbits, orc_cf *sym)
{
	if (nbits % 3)
		return -1;
	for (int i = 0; i < nbits / 3; i++) {
		unsigned idx = ((unsigned)(bits[3 * i + 0] & 1) << 0) | ((unsigned)(bits[3 * i + 1] & 1) << 1) |
			       ((unsigned)(bits[3 * i + 2] & 1) << 2);
		sym[i] = cf((float)PSK8[idx][0], (float)PSK8[idx][1]);
	}
	return nbits / 3;
}

/* sigProcLib.cpp:917-936 with empty=false -> shapeEdgeBurst :739-763 */
int orc_modulate_edge_burst(const uint8_t *bits, int nbits, orc_cf *out)
{
	if (!T_ready) orc_setup();
	orc_cf sym[160];
	if (nbits > 468)
		return -1;
	int nsyms = map_edge_symbols(bits, nbits, sym);
	if (nsyms < 0)
		return -1;
	const int nsamps = 625, sps = 4;
	if (nsyms * sps > nsamps)
		nsyms = 156;
	orc_cf *b = calloc(nsamps + 8, sizeof(orc_cf));
	for (int i = 0; i < nsyms; i++) {
		/* float phase = i * 3.0f * M_PI / 8.0f;  cos(float) -> float overload */
		float phase = (float)((double)((float)i * 3.0f) * M_PI / 8.0f);
		orc_cf rot = cf(cosf(phase), sinf(phase));
		b[sps + i * sps] = cmul(sym[i], rot);
	}
	conv_span_rtaps(b, nsamps, T.pulse4_c0, 16, out, 0, nsamps, 1);
	free(b);
	return nsamps;
}

/* sigProcLib.cpp:1100-1118 */
static orc_cf interpolate_point(const orc_cf *sig, int size, float ix)
{
	int start = (int)(floorf(ix) - 10);
	if (start < 0) start = 0;
	int end = (int)(floorf(ix) + 11);
	if ((size_t)(unsigned)end > (size_t)size - 1) end = size - 1;

	orc_cf p = cf(0.0f, 0.0f);
	for (int i = start; i < end; i++) {
		orc_cf t = cmulr(sig[i], sinc_lut(M_PI_F * (i - ix)));
		p.re += t.re;
		p.im += t.im;
	}
	return p;
}

/* sigProcLib.cpp:1120-1139 */
static orc_cf fast_peak_detect(const orc_cf *sig, int size, float *index)
{
	float max = 0.0f;
	orc_cf amp = cf(0.0f, 0.0f);
	int idx = -1;
	for (int i = 0; i < size; i++) {
		float val = cnorm2(sig[i]);
		if (val > max) {
			max = val;
			idx = i;
			amp = sig[i];
		}
	}
	if (index) *index = (float)idx;
	return amp;
}

/* sigProcLib.cpp:1141-1186 */
static orc_cf peak_detect(const orc_cf *sig, int size, float *peak_index, float *avg_pwr)
{
	float max_val = 0.0f;           /* maxVal.real() */
	float max_index = -1;
	float sum_power = 0.0f;

	for (int i = 0; i < size; i++) {
		float p = cnorm2(sig[i]);
		if (p > max_val) {
			max_val = p;
			max_index = i;
		}
		sum_power += p;
	}

	float early = max_index - 1;
	float late = max_index + 1;
	float incr = 0.5;
	while (incr > 1.0 / 1024.0) {
		orc_cf ep = interpolate_point(sig, size, early);
		orc_cf lp = interpolate_point(sig, size, late);
		if (cnorm2(ep) < cnorm2(lp))            /* Complex.h:102 operator< on norm2 */
			early += incr;
		else if (cnorm2(ep) > cnorm2(lp))
			early -= incr;
		else
			break;
		incr /= 2.0;
		late = early + 2.0;
	}

	max_index = early + 1.0;
	orc_cf mv = interpolate_point(sig, size, max_index);

	if (peak_index) *peak_index = max_index;
	if (avg_pwr) *avg_pwr = (sum_power - cnorm2(mv)) / (size - 1);
	return mv;
}

/* Shared tail of generateMidamble/generateDummyMidamble/generateRACHSequence/generateSCHSequence:
 * autocorr = convolve(shaped, seq, NULL, NO_DELAY); gain = peakDetect(autocorr,&toa); toa -= off */
static void finish_corr_seq(orc_corr_seq *cs, const orc_cf *shaped, int shaped_len, double toa_off)
{
	orc_cf ac[160];
	float toa;
	conv_span(shaped, shaped_len, cs->seq, cs->n, 0, ac, cs->n / 2, shaped_len, 1);
	cs->gain = peak_detect(ac, shaped_len, &toa, NULL);
	cs->toa = (float)(toa - toa_off);
}

/* sigProcLib.cpp:1227-1299 (sps = 1) and :1301-1370 (dummy) */
static void generate_midamble(orc_corr_seq *cs, const char *tsc_bits)
{
	uint8_t bits[32];
	orc_cf mid[32], shaped[32];
	int n = bits_from_str(tsc_bits, bits);                          /* 26 */

	int mlen = rotate_burst(bits + 5, 16, 0, 1, mid);                /* segment(5,16), emptyPulse */
	int slen = modulate_burst_basic(bits, n, 0, 1, shaped);

	for (int i = 0; i < mlen; i++) mid[i] = cmul(mid[i], cf(-1.0f, 0.0f));      /* scaleVector :1257 */
	for (int i = 0; i < slen; i++) shaped[i] = cmul(shaped[i], cf(0.0f, 1.0f)); /* :1258 */
	for (int i = 0; i < mlen; i++) mid[i] = cf(mid[i].re, -mid[i].im);          /* conjugateVector :1260 */

	cs->n = mlen;
	memset(cs->seq, 0, sizeof(cs->seq));
	memcpy(cs->seq, mid, (size_t)mlen * sizeof(orc_cf));
	finish_corr_seq(cs, shaped, slen, 13.5);                         /* :1268-1283 */
}

/* sigProcLib.cpp:1405-1465 (sps = 1); SCH :1467-1527 uses the full sequence and 32.5 */
static void generate_sync_seq(orc_corr_seq *cs, const char *bitstr, int corr_bits, double toa_off)
{
	uint8_t bits[80];
	orc_cf seq0[80], seq1[80];
	int n = bits_from_str(bitstr, bits);

	int l0 = modulate_burst_basic(bits, n, 0, 1, seq0);
	int l1 = rotate_burst(bits, corr_bits, 0, 1, seq1);
	for (int i = 0; i < l1; i++) seq1[i] = cf(seq1[i].re, -seq1[i].im);

	cs->n = l1;
	memset(cs->seq, 0, sizeof(cs->seq));
	memcpy(cs->seq, seq1, (size_t)l1 * sizeof(orc_cf));
	finish_corr_seq(cs, seq0, l0, toa_off);
}

/* sigProcLib.cpp:1372-1403 */
static void generate_edge_midamble(orc_corr_seq *cs, const char *bitstr)
{
	uint8_t bits[80];
	orc_cf sym[16];
	bits_from_str(bitstr, bits);
	int nsyms = map_edge_symbols(bits + 15, 48, sym);               /* segment(15,48) */
	memset(cs->seq, 0, sizeof(cs->seq));
	for (int i = 0; i < nsyms; i++) {
		/* rotateEdgeBurst :672-689, sps=1: float phase = i*3.0f*M_PI/8.0f; cos(float) */
		float phase = (float)((double)((float)i * 3.0f) * M_PI / 8.0f);
		orc_cf rot = cf(cosf(phase), sinf(phase));
		orc_cf v = cmul(sym[i], rot);
		cs->seq[i] = cf(v.re, -v.im);                                /* conjugateVector */
	}
	cs->n = nsyms;
	/* :1397  Complex<float>(-19.6432, 19.5006) / 1.18  -> operator/(Real): (r/a, i/a), a = float(1.18) */
	cs->gain = cf((float)-19.6432 / (float)1.18, (float)19.5006 / (float)1.18);
	cs->toa = 0;
}

/* sigProcLib.cpp:1005-1044 */
static void generate_delay_filters(void)
{
	const int h_len = ORC_DELAY_HLEN;
	float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;

	for (int i = 0; i < ORC_DELAYFILTS; i++) {
		float *h = T.delay_filt[i];
		float sum = 0.0f;
		for (int n = 0; n < h_len; n++) {
			float k = (float)n;
			/* M_PI_F * (k - (float)h_len/2.0 - (float)i/DELAYFILTS): double, then float arg */
			float arg = (float)(M_PI_F * (k - (float)h_len / 2.0 - (float)i / ORC_DELAYFILTS));
			float v = sinc_lut(arg);
			/* *itr *= (double expr): Complex::operator*=(Real) takes it rounded to float */
			float w = (float)(a0 -
				a1 * cos(2 * M_PI * n / (h_len - 1)) +
				a2 * cos(4 * M_PI * n / (h_len - 1)) -
				a3 * cos(6 * M_PI * n / (h_len - 1)));
			v *= w;
			h[h_len - 1 - n] = v;           /* *--itr, from end() */
			sum += v;
		}
		for (int n = 0; n < h_len; n++)
			h[n] /= sum;
	}
}

/* ------------------------------------------------------------------------------------------
 * Resampler.cpp
 * ---------------------------------------------------------------------------------------- */
struct orc_resampler {
	int p, q, filt_len;
	float *part;         /* p x filt_len real taps, reversed */
};

/* Resampler.cpp:39-45 */
static float rs_sinc(float x)
{
	if (x == 0.0)
		return 0.9999999999;
	return sin(M_PI * x) / (M_PI * x);
}

/* Resampler.cpp:47-96 + :152-168 */
orc_resampler *orc_resampler_new(int p, int q, int filt_len, float bw)
{
	if (p <= 0 || q <= 0 || filt_len <= 0)
		return NULL;
	orc_resampler *r = calloc(1, sizeof(*r));
	r->p = p; r->q = q; r->filt_len = filt_len;
	size_t plen = (size_t)p * filt_len;
	float *proto = calloc(plen, sizeof(float));
	r->part = calloc(plen, sizeof(float));

	float sum = 0.0f, scale = 0.0f;
	float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
	float cutoff = (p > q) ? (float)p : (float)q;
	float midpt = (plen - 1) / 2.0;
	for (size_t i = 0; i < plen; i++) {
		proto[i] = rs_sinc(((float)i - midpt) / cutoff * bw);
		proto[i] *= a0 -
			    a1 * cos(2 * M_PI * i / (plen - 1)) +
			    a2 * cos(4 * M_PI * i / (plen - 1)) -
			    a3 * cos(6 * M_PI * i / (plen - 1));
		sum += proto[i];
	}
	scale = p / sum;
	for (int i = 0; i < filt_len; i++)
		for (int n = 0; n < p; n++)
			r->part[(size_t)n * filt_len + (filt_len - 1 - i)] = proto[(size_t)i * p + n] * scale;
	free(proto);
	return r;
}

void orc_resampler_free(orc_resampler *r)
{
	if (r) { free(r->part); free(r); }
}

const float *orc_resampler_partition(const orc_resampler *r, int path)
{
	return &r->part[(size_t)path * r->filt_len];
}

/* Resampler.cpp:131-150: out[i] = sum_k in[n - (L-1) + k] * part[path][k], n=(q*i)/p, path=(q*i)%p */
/* Resampler::rotate over plugged arch kernels: one convolve_real() call per output, len 1, start n, straight on the caller's
 * buffer as the reference does (Resampler.cpp:143-147; partitions are memalign'ed there: aligned taps here) */
static int resampler_rotate_arch(const orc_resampler *r, const orc_cf *in, int in_len, orc_cf *out, int out_len)
{
	float *hp = NULL;
	const int L = r->filt_len;
	if (posix_memalign((void **)&hp, 16, (size_t)r->p * ((size_t)L + 4) * 2 * sizeof(float)))
		return -1;
	for (int path = 0; path < r->p; path++)
		for (int k = 0; k < L; k++) {
			hp[((size_t)path * (L + 4) + k) * 2] = r->part[(size_t)path * L + k];
			hp[((size_t)path * (L + 4) + k) * 2 + 1] = 0.0f;
		}
	for (int i = 0; i < out_len; i++) {
		int n = (r->q * i) / r->p;
		int path = (r->q * i) % r->p;
		orc_cf y = cf(0.0f, 0.0f);
		A_real((const float *)(in - L), in_len + L, &hp[(size_t)path * (L + 4) * 2], L, (float *)&y, out_len - i, n + L, 1);
		out[i] = y;
	}
	free(hp);
	return out_len;
}

int orc_resampler_rotate(const orc_resampler *r, const orc_cf *in, int in_len, orc_cf *out, int out_len)
{
	if (A_real && (r->filt_len % 4) == 0)
		return resampler_rotate_arch(r, in, in_len, out, out_len);
	for (int i = 0; i < out_len; i++) {
		int n = (r->q * i) / r->p;
		int path = (r->q * i) % r->p;
		const float *h = &r->part[(size_t)path * r->filt_len];
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < r->filt_len; k++) {
			const orc_cf xv = in[n - (r->filt_len - 1) + k];
			yr += xv.re * h[k];
			yi += xv.im * h[k];
		}
		out[i] = cf(yr, yi);
	}
	return out_len;
}

/* ------------------------------------------------------------------------------------------
 * sigProcLibSetup, sigProcLib.cpp:2139-2172
 * ---------------------------------------------------------------------------------------- */
int orc_setup(void)
{
	if (T_ready)
		return 1;
	memset(&T, 0, sizeof(T));
	generate_sinc_table();
	init_gmsk_rotation_tables();
	generate_gsm_pulses();
	T_ready = 1;    /* modulators below need the tables above */

	for (int i = 0; i < 3; i++)
		generate_sync_seq(&T.rach[i], RACH_BITS[i], 40, 20.5);
	generate_sync_seq(&T.sch, SCH_BITS, 64, 32.5);
	generate_midamble(&T.dummy, DUMMY_TSC_BITS);
	for (int tsc = 0; tsc < 8; tsc++) {
		generate_midamble(&T.midamble[tsc], TSC_BITS[tsc]);
		generate_edge_midamble(&T.edge_midamble[tsc], EDGE_TSC_BITS[tsc]);
	}
	generate_delay_filters();

	orc_resampler *dn = orc_resampler_new(1, 4, 16, 1.0f);
	memcpy(T.dec_taps, orc_resampler_partition(dn, 0), sizeof(T.dec_taps));
	orc_resampler_free(dn);
	return 1;
}

const orc_tables *orc_get_tables(void)
{
	if (!T_ready) orc_setup();
	return &T;
}

/* ------------------------------------------------------------------------------------------
 * Detection
 * ---------------------------------------------------------------------------------------- */
/* sigProcLib.cpp:1573-1585 */
float orc_energy_detect(const orc_cf *burst, int n, unsigned window)
{
	float energy = 0.0f;
	if (window == 0) return 0.0f;
	if (window > (unsigned)n) window = n;
	for (unsigned i = 0; i < window; i++)
		energy += cnorm2(burst[4 * i]);      /* windowItr += 4 regardless of sps */
	return energy / window;
}

/* sigProcLib.cpp:546-556 */
void orc_vector_slicer(float *dest, const float *src, size_t len)
{
	for (size_t i = 0; i < len; i++) {
		dest[i] = 0.5 * (src[i] + 1.0f);
		if (dest[i] > 1.0)
			dest[i] = 1.0;
		else if (dest[i] < 0.0)
			dest[i] = 0.0;
	}
}

/* sigProcLib.cpp:1587-1601 downsampleBurst + Resampler::rotate (P=1, Q=4):
 * in = 16 zeros + burst[0..in_len); out[i] = sum_k in[4i - 15 + k] * g[k] */
static void downsample_burst(const orc_cf *burst, int in_len, orc_cf *out, int out_len)
{
	if (A_real) {
		/* signalVector in(in_len, dnsampler->len()): 16 samples of zero head-room, then Resampler::rotate() (:1590-1598):
		 * one convolve_real() call per output, len 1, on that buffer */
		float *buf = NULL, h2[2 * 16] __attribute__((aligned(16)));
		if (posix_memalign((void **)&buf, 16, ((size_t)in_len + 16 + 4) * sizeof(orc_cf)))
			return;
		memset(buf, 0, ((size_t)in_len + 16 + 4) * sizeof(orc_cf));
		memcpy(buf + 32, burst, (size_t)in_len * sizeof(orc_cf));
		for (int k = 0; k < 16; k++) { h2[2 * k] = T.dec_taps[k]; h2[2 * k + 1] = 0.0f; }
		for (int i = 0; i < out_len; i++) {
			orc_cf y = cf(0.0f, 0.0f);
			A_real(buf, in_len + 16, h2, 16, (float *)&y, out_len - i, 4 * i + 16, 1);
			out[i] = y;
		}
		free(buf);
		return;
	}
	for (int i = 0; i < out_len; i++) {
		float yr = 0.0f, yi = 0.0f;
		for (int k = 0; k < 16; k++) {
			int j = 4 * i - 15 + k;
			orc_cf xv = (j >= 0 && j < in_len) ? burst[j] : cf(0.0f, 0.0f);
			yr += xv.re * T.dec_taps[k];
			yi += xv.im * T.dec_taps[k];
		}
		out[i] = cf(yr, yi);
	}
}

/* sigProcLib.cpp:1541-1571 */
static float compute_peak_ratio(const orc_cf *corr, int len, int sps, float toa, orc_cf amp)
{
	int num = 0;
	float rms, avg = 0.0f;

	if ((toa < 0.0) || (toa > (float)len))
		return 0.0f;
	int peak = (int)rintf(toa);
	for (int i = 2 * sps; i <= 5 * sps; i++) {
		if (peak - i >= 0) {
			avg += cnorm2(corr[peak - i]);
			num++;
		}
		if (peak + i < len) {
			avg += cnorm2(corr[peak + i]);
			num++;
		}
	}
	if (num < 5)
		return 0.0f;
	rms = sqrtf(avg / (float)num) + 0.00001;
	return cabs_(amp) / rms;
}

/* sigProcLib.cpp:1608-1639 */
static float compute_ci(const orc_cf *burst, int burst_size, const orc_corr_seq *sync,
			float toa, int start, orc_cf xcorr)
{
	const int N = sync->n;
	float S, C;
	const int ps = start + 1 - N + (int)roundf(toa);

	if (ps < 0)
		return 0;
	if (ps + N > burst_size)
		return 0;

	S = 0.0f;
	for (int i = 0, j = ps; i < N; i++, j++)
		S += cnorm2(burst[j]);
	S /= N;

	C = cnorm2(xcorr) / ((N - 1) * cabs_(sync->gain));
	return 3.0103f * log2f(C / (S - C));
}

/* sigProcLib.cpp:1672-1708: detectBurst() from the correlation onwards (corr_in is at 1 SPS) */
static int detect_burst_1sps(const orc_cf *corr_in, int corr_in_len, orc_cf *corr, const orc_corr_seq *sync,
			     float thresh, int start, int len, orc_ebp *ebp)
{
	const int sps = 1;
	orc_cf xcorr;

	/* Correlate :1674 (CUSTOM span) */
	conv_span(corr_in, corr_in_len, sync->seq, sync->n, 0, corr, start, len, 1);

	ebp->amp = fast_peak_detect(corr, len, &ebp->toa);

	if ((ebp->toa < 3 * sps) || (ebp->toa > len - 3 * sps))
		return 0;

	if (compute_peak_ratio(corr, len, sps, ebp->toa, ebp->amp) < thresh)
		return 0;

	xcorr = peak_detect(corr, len, &ebp->toa, NULL);
	ebp->ci = compute_ci(corr_in, corr_in_len, sync, ebp->toa, start, xcorr);
	ebp->amp = cdiv(xcorr, sync->gain);
	ebp->toa = ebp->toa - sync->toa;
	return 1;
}

/* sigProcLib.cpp:1649-1709 */
static int detect_burst(const orc_cf *burst, int n, orc_cf *corr, const orc_corr_seq *sync,
			float thresh, int sps, int start, int len, orc_ebp *ebp)
{
	orc_cf dec[160];

	switch (sps) {
	case 1:
		return detect_burst_1sps(burst, n, corr, sync, thresh, start, len, ebp);
	case 4:
		downsample_burst(burst, 624, dec, 156);
		return detect_burst_1sps(dec, 156, corr, sync, thresh, start, len, ebp);
	default:
		return -1;
	}
}

/* sigProcLib.cpp:1805-1861 detectSCHBurst.  The burst is always decimated by 4 here (:1841), whatever `sps`
 * says, and only its first 4*len samples are used; n < 4*len trips the reference's copyToSegment assertion
 * (Vector.h:236-237) and returns -1 here. */
int orc_detect_sch_burst(const orc_cf *burst, int n, float thresh, int sps, int state, orc_ebp *ebp)
{
	if (!T_ready) orc_setup();
	int rc, start, target, head, tail, len;

	if ((sps != 1) && (sps != 4))
		return -1;

	target = 3 + 39 + 64;
	switch (state) {
	case ORC_SCH_DETECT_NARROW:
		head = 4;
		tail = 4;
		break;
	case ORC_SCH_DETECT_BUFFER:
		target = 1;
		head = 0;
		tail = (12 * 8 * 625) / 4;
		break;
	case ORC_SCH_DETECT_FULL:
	default:
		head = target - 1;
		tail = 39 + 3 + 9;
		break;
	}
	start = (target - head) * 1 - 1;
	len = (head + tail) * 1;
	if (n < len * 4)
		return -1;

	orc_cf *corr = malloc(2 * (size_t)len * sizeof(orc_cf)), *dec = corr + len;
	downsample_burst(burst, len * 4, dec, len);
	rc = detect_burst_1sps(dec, len, corr, &T.sch, thresh, start, len, ebp);
	free(corr);

	if (rc < 0) {
		return -1;
	} else if (!rc) {
		ebp->amp = cf(0.0f, 0.0f);
		ebp->toa = 0.0f;
		return 0;
	}
	if (state == ORC_SCH_DETECT_BUFFER)
		ebp->toa = ebp->toa - (3 + 39 + 64);
	else
		ebp->toa = ebp->toa - head;
	return rc;
}

/* sigProcLib.cpp:1711-1722 */
static float max_amplitude(const orc_cf *burst, int n)
{
	float max = 0.0f;
	for (int i = 0; i < n; i++) {
		if (fabsf(burst[i].re) > max) max = fabsf(burst[i].re);
		if (fabsf(burst[i].im) > max) max = fabsf(burst[i].im);
	}
	return max;
}

/* sigProcLib.cpp:1732-1771 */
static int detect_general_burst(const orc_cf *burst, int n, float thresh, int sps,
				int target, int head, int tail,
				const orc_corr_seq *sync, orc_ebp *ebp)
{
	int rc, start, len;
	int clipping = 0;

	if ((sps != 1) && (sps != 4))
		return -ORC_SIGERR_UNSUPPORTED;

	if (max_amplitude(burst, n) > CLIP_THRESH)
		clipping = 1;

	start = target - head - 1;
	len = head + tail;
	orc_cf *corr = calloc((size_t)len, sizeof(orc_cf));

	rc = detect_burst(burst, n, corr, sync, thresh, sps, start, len, ebp);
	free(corr);
	if (rc < 0) {
		return -ORC_SIGERR_INTERNAL;
	} else if (!rc) {
		ebp->amp = cf(0.0f, 0.0f);
		ebp->toa = 0.0f;
		ebp->ci = 0.0f;
		return clipping ? -ORC_SIGERR_CLIP : ORC_SIGERR_NONE;
	}

	ebp->toa -= head;
	return 1;
}

/* sigProcLib.cpp:1782-1803 */
static int detect_rach_burst(const orc_cf *burst, int n, float threshold, int sps,
			     unsigned max_toa, int ext, orc_ebp *ebp)
{
	int rc = 0, target = 8 + 40, head = 8, tail = 8 + max_toa;
	int num_seq = ext ? 3 : 1;

	for (int i = 0; i < num_seq; i++) {
		rc = detect_general_burst(burst, n, threshold, sps, target, head, tail, &T.rach[i], ebp);
		if (rc > 0) {
			ebp->tsc = i;
			break;
		}
	}
	return rc;
}

/* sigProcLib.cpp:1863-1877 */
static int detect_dummy_burst(const orc_cf *burst, int n, float threshold, int sps,
			      unsigned max_toa, orc_ebp *ebp)
{
	int target = 3 + 58 + 16 + 5, head = 10, tail = 6 + max_toa;
	ebp->tsc = 0;
	return detect_general_burst(burst, n, threshold, sps, target, head, tail, &T.dummy, ebp);
}

/* sigProcLib.cpp:1887-1904 */
static int analyze_traffic_burst(const orc_cf *burst, int n, unsigned tsc, float threshold,
				 int sps, unsigned max_toa, orc_ebp *ebp)
{
	if (tsc > 7)
		return -ORC_SIGERR_UNSUPPORTED;
	int target = 3 + 58 + 16 + 5, head = 10, tail = 6 + max_toa;
	ebp->tsc = tsc;
	return detect_general_burst(burst, n, threshold, sps, target, head, tail, &T.midamble[tsc], ebp);
}

/* sigProcLib.cpp:1906-1924 */
static int detect_edge_burst(const orc_cf *burst, int n, unsigned tsc, float threshold,
			     int sps, unsigned max_toa, orc_ebp *ebp)
{
	if (tsc > 7)
		return -ORC_SIGERR_UNSUPPORTED;
	int target = 3 + 58 + 16 + 5, head = 6, tail = 6 + max_toa;
	ebp->tsc = tsc;
	return detect_general_burst(burst, n, threshold, sps, target, head, tail, &T.edge_midamble[tsc], ebp);
}

/* sigProcLib.cpp:1926-1957 */
int orc_detect_any_burst(const orc_cf *burst, int n, unsigned tsc, float threshold, int sps,
			 int type, unsigned max_toa, orc_ebp *ebp)
{
	int rc = 0;
	if (!T_ready) orc_setup();

	switch (type) {
	case ORC_EDGE:
		rc = detect_edge_burst(burst, n, tsc, threshold, sps, max_toa, ebp);
		if (rc > 0)
			break;
		else
			type = ORC_TSC;
		/* fall through */
	case ORC_TSC:
		rc = analyze_traffic_burst(burst, n, tsc, threshold, sps, max_toa, ebp);
		break;
	case ORC_EXT_RACH:
	case ORC_RACH:
		rc = detect_rach_burst(burst, n, threshold, sps, max_toa, type == ORC_EXT_RACH, ebp);
		break;
	case ORC_IDLE:
		rc = detect_dummy_burst(burst, n, threshold, sps, max_toa, ebp);
		break;
	default:
		break;
	}

	if (rc > 0)
		return type;
	return rc;
}

/* ------------------------------------------------------------------------------------------
 * Demodulation
 * ---------------------------------------------------------------------------------------- */
/* sigProcLib.cpp:1046-1098 */
void orc_delay_vector(const orc_cf *in, int n, float delay, orc_cf *out)
{
	if (!T_ready) orc_setup();
	int whole = (int)floorf(delay);
	float frac = delay - whole;
	orc_cf *shift = malloc((size_t)n * sizeof(orc_cf));

	if ((double)fabsf(frac) > 1e-2) {
		int index = (int)floorf(frac * (float)ORC_DELAYFILTS);
		/* convolve(in, h, NULL, NO_DELAY): start = h_len/2 = 10 */
		conv_span_rtaps(in, n, T.delay_filt[index], ORC_DELAY_HLEN, shift, ORC_DELAY_HLEN / 2, n, 1);   /* aligned (:1021) */
	} else {
		memcpy(shift, in, (size_t)n * sizeof(orc_cf));
	}

	/* Integer sample shift :1071-1090 */
	for (int i = 0; i < n; i++) {
		int j = i - whole;      /* whole<0: left shift by -whole; whole>=0: right shift */
		out[i] = (j >= 0 && j < n) ? shift[j] : cf(0.0f, 0.0f);
	}
	free(shift);
}

/* sigProcLib.cpp:2030-2048 demodCommon; returns output length */
static int demod_common(const orc_cf *burst, int n, int sps, const orc_ebp *ebp, orc_cf *out)
{
	if ((sps != 1) && (sps != 4))
		return -1;
	orc_cf *delay = malloc((size_t)n * sizeof(orc_cf));
	orc_delay_vector(burst, n, -ebp->toa * (float)sps, delay);
	orc_cf scale = cdiv(cf(1.0f, 0.0f), ebp->amp);         /* (complex) 1.0 / ebp->amp */
	for (int i = 0; i < n; i++)
		delay[i] = cmul(delay[i], scale);                   /* scaleVector :1198-1205 */

	int olen;
	if (sps == 1) {
		memcpy(out, delay, (size_t)n * sizeof(orc_cf));
		olen = n;
	} else {
		downsample_burst(delay, 624, out, 156);
		olen = 156;
	}
	free(delay);
	return olen;
}

/* sigProcLib.cpp:2055-2072 */
static int demod_gmsk_burst(const orc_cf *burst, int n, int sps, const orc_ebp *ebp, float *soft)
{
	orc_cf *dec = malloc((size_t)(n > 160 ? n : 160) * sizeof(orc_cf));
	int olen = demod_common(burst, n, sps, ebp, dec);
	if (olen < 0) { free(dec); return -1; }
	for (int i = 0; i < olen; i++) {
		orc_cf v = cmul(T.rrot1[i], dec[i]);              /* GMSKReverseRotate(*dec, 1) */
		soft[i] = v.re;                                     /* signalToSoftVector */
	}
	free(dec);
	return olen;
}

/* sigProcLib.cpp:582-588 */
static void rotate_burst2(orc_cf *b, int n, double phase)
{
	orc_cf rot = cf((float)cos(phase), (float)sin(phase));
	for (int i = 0; i < n; i++)
		b[i] = cmul(b[i], rot);
}

/* sigProcLib.cpp:2074-2093 */
static float compute_edge_ci(const orc_cf *rot, int n)
{
	float err_pwr = 0.0f;
	float step = 2.0f * M_PI_F / 8.0f;
	for (int i = 8; i < n - 8; i++) {
		orc_cf sym = rot[i];
		/* sym.arg() = ::atan2(i, r) float overload */
		float phase = step * roundf(atan2f(sym.im, sym.re) / step);
		orc_cf ideal = cf(cosf(phase), sinf(phase));
		orc_cf err = cf(ideal.re - sym.re, ideal.im - sym.im);
		err_pwr += cnorm2(err);
	}
	return 3.0103f * log2f(1.0f * (n - 16) / err_pwr);
}

/* sigProcLib.cpp:2105-2128 (+ derotateEdgeBurst :691-711, softSliceEdgeBurst :1962-2006) */
static int demod_edge_burst(const orc_cf *burst, int n, int sps, orc_ebp *ebp, float *soft)
{
	const int cap = n > 160 ? n : 160;
	orc_cf *dec = malloc(3 * (size_t)cap * sizeof(orc_cf)), *eq = dec + cap, *rot = eq + cap;
	int olen = demod_common(burst, n, sps, ebp, dec);
	if (olen < 0) { free(dec); return -1; }

	/* eq = convolve(dec, c0_inv, NULL, NO_DELAY): 5 real taps, start = 2 */
	conv_span_rtaps(dec, olen, T.c0_inv, 5, eq, 2, olen, 0);   /* c0_inv->setAligned(false) (:412): base_convolve_real */
	for (int i = 0; i < olen; i++) {
		float phase = (float)((double)((float)(i % 16) * 3.0f) * M_PI / 8.0f);
		orc_cf r = cf(cosf(phase), -sinf(phase));
		rot[i] = cmul(eq[i], r);
	}
	ebp->ci = compute_edge_ci(rot, olen);

	const int nsyms = 148;
	if (olen < nsyms) { free(dec); return -1; }
	rotate_burst2(rot, olen, -M_PI / 8.0);
	for (int i = 0; i < nsyms; i++) {
		soft[3 * i + 0] = -rot[i].im;
		soft[3 * i + 1] = rot[i].re;
	}
	for (int i = 0; i < olen; i++)
		rot[i] = cf(fabsf(rot[i].re), fabsf(rot[i].im));
	rotate_burst2(rot, olen, -M_PI / 4.0);
	for (int i = 0; i < nsyms; i++)
		soft[3 * i + 2] = -rot[i].im;
	free(dec);
	return nsyms * 3;
}

/* sigProcLib.cpp:2130-2137 */
int orc_demod_any_burst(const orc_cf *burst, int n, int type, int sps, orc_ebp *ebp, float *soft)
{
	if (!T_ready) orc_setup();
	if (type == ORC_EDGE)
		return demod_edge_burst(burst, n, sps, ebp, soft);
	else
		return demod_gmsk_burst(burst, n, sps, ebp, soft);
}

/* ------------------------------------------------------------------------------------------
 * Batched DSP core of Transceiver::pullRadioVector, Transceiver.cpp:665-815
 * (FIFO, clock, noise history and counters stay with the caller)
 * ---------------------------------------------------------------------------------------- */
void orc_pull_batch(const int16_t *iq, size_t n_bursts, int burst_len, int sps,
		    const orc_burst_params *params, float threshold, double full_scale,
		    orc_burst_result *res, float *soft, int soft_stride, int slice)
{
	if (!T_ready) orc_setup();
	orc_cf *burst = malloc((size_t)burst_len * sizeof(orc_cf));
	float raw[448];

	for (size_t b = 0; b < n_bursts; b++) {
		const orc_burst_params *p = &params[b];
		orc_burst_result *r = &res[b];
		float *so = soft ? &soft[b * (size_t)soft_stride] : NULL;
		memset(r, 0, sizeof(*r));
		if (so) memset(so, 0, (size_t)soft_stride * sizeof(float));
		r->idle = 1;

		if (p->type == ORC_OFF)                       /* :704-707 */
			continue;

		/* radioInterface.cpp:344-348 convert_short_float, no scaling */
		orc_convert_short_float((float *)burst, &iq[b * (size_t)burst_len * 2], burst_len * 2);

		/* :724-746 (one diversity path) */
		float pow = orc_energy_detect(burst, burst_len, 20 * sps);
		float avg = 0.0f;
		avg += pow;
		avg = sqrtf(avg / 1);
		r->energy = pow;
		r->rssi = (float)(20.0 * log10(full_scale / avg));      /* :751 without rssi_offset */
		r->clip = max_amplitude(burst, burst_len) > CLIP_THRESH;

		if (p->type == ORC_IDLE)                      /* :754-755 */
			continue;

		orc_ebp ebp;
		memset(&ebp, 0, sizeof(ebp));
		int rc = orc_detect_any_burst(burst, burst_len, p->tsc, threshold, sps, p->type, p->max_toa, &ebp);
		r->rc = rc;
		if (rc <= 0)                                  /* :769-782 */
			continue;

		int nsoft = orc_demod_any_burst(burst, burst_len, rc, sps, &ebp, raw);
		r->toa = ebp.toa;
		r->amp_re = ebp.amp.re;
		r->amp_im = ebp.amp.im;
		r->ci = ebp.ci;
		r->tsc = ebp.tsc;
		r->idle = 0;
		int nbits = (nsoft == 444) ? 444 : 148;       /* :793-800 */
		r->nbits_div4 = (uint8_t)(nbits / 4);
		if (so) {
			if (slice) {
				int m = nbits < soft_stride ? nbits : soft_stride;
				orc_vector_slicer(so, raw, (size_t)m);      /* :803 */
			} else {
				int m = nsoft < soft_stride ? nsoft : soft_stride;
				memcpy(so, raw, (size_t)m * sizeof(float));
			}
		}
	}
	free(burst);
}

/* ----------------------------------------------------------------------------------------
 * pullRadioVector() with diversity paths (Transceiver.cpp:669-671, :723-751): the burst arrives on n_paths receive
 * paths; each path's energy is measured, the first path with the highest energy is the one detected and demodulated,
 * and the power levels come from the path average:
 *     float max = -1.0, avg = 0.0; int max_i = -1;
 *     for (i < chans()) { pow = energyDetect(path i, 20 * sps); if (pow > max) { max = pow; max_i = i; } avg += pow; }
 *     avg = sqrt(avg / chans());   rssi = 20 log10(rxFullScale / avg)
 * iq: n_bursts x n_paths x burst_len x 2 int16.  res[b].energy = avg * avg's argument (sum pow / chans), path[b] = max_i.
 * ---------------------------------------------------------------------------------------- */
void orc_pull_batch_div(const int16_t *iq, size_t n_bursts, int n_paths, int burst_len, int sps,
			const orc_burst_params *params, float threshold, double full_scale,
			orc_burst_result *res, float *soft, int soft_stride, int slice, uint8_t *path)
{
	orc_cf *burst = malloc((size_t)burst_len * sizeof(orc_cf));
	for (size_t b = 0; b < n_bursts; b++) {
		const int16_t *paths = &iq[b * (size_t)n_paths * burst_len * 2];
		float max = -1.0f, avg = 0.0f;
		int max_i = -1;
		for (int i = 0; i < n_paths; i++) {
			orc_convert_short_float((float *)burst, &paths[(size_t)i * burst_len * 2], burst_len * 2);
			float pow = orc_energy_detect(burst, burst_len, 20 * sps);
			if (pow > max) {
				max = pow;
				max_i = i;
			}
			avg += pow;
		}
		if (max_i < 0) max_i = 0;                         /* "Received empty burst": not reachable with finite samples */
		if (path) path[b] = (uint8_t)max_i;
		orc_pull_batch(&paths[(size_t)max_i * burst_len * 2], 1, burst_len, sps, &params[b], threshold, full_scale,
			       &res[b], soft ? &soft[b * (size_t)soft_stride] : NULL, soft_stride, slice);
		if (params[b].type == ORC_OFF)
			continue;
		res[b].energy = avg / (float)n_paths;
		avg = sqrtf(avg / (float)n_paths);                /* :741 */
		res[b].rssi = (float)(20.0 * log10(full_scale / avg));   /* :751 without rssi_offset */
	}
	free(burst);
}

/* proto_trxd.c:36-45 */
int orc_trxd_toa256(double toa)
{
	return (int)(toa * 256.0 + 0.5);
}

/* proto_trxd.c:47-52 */
int16_t orc_trxd_ci_cb(float ci)
{
	return (int16_t)((ci * 10) + 0.5);
}

/* proto_trxd.c:61-66 */
void orc_trxd_soft_u8(uint8_t *dst, const float *rx_burst, unsigned nbits)
{
	for (unsigned i = 0; i < nbits; i++)
		dst[i] = (uint8_t)round(rx_burst[i] * 255.0);
}

/* proto_trxd.c:68-117, trxd_send_burst_ind_v0() / _v1(): the datagram handed to write(), built field by field
 * as trxd_fill_common() :28-34, trxd_fill_v0_specific() :36-45, trxd_fill_v1_specific() :47-60 and
 * trxd_fill_burst_normalized255() :62-66 do, with the packed little-endian bit-field layouts of proto_trxd.h:56-106
 * written out (tn:3 | reserved:1 | version:4;  tsc:3 | modulation:4 | idle:1).
 * Returns the datagram length; 0 when nothing is sent (v0 and idle, :71-73); -1 for an unknown version.
 * Two places where the reference's bytes are not defined and this restatement picks a value:
 *   - v0 trailing byte soft_bits[nbits] is uninitialised stack in the reference (:83-87): written as 0 here;
 *   - `v0->rssi = bi->rssi` converts double to uint8_t, undefined outside 0..255: saturated here. */
int orc_trxd_pack(uint8_t *buf, unsigned version, uint32_t fn, uint8_t tn, double rssi, double toa, int idle,
		  int modulation_8psk, uint8_t tss, uint8_t tsc, float ci, const float *rx_burst, unsigned nbits)
{
	if (version > 1)
		return -1;
	if (version == 0 && idle)
		return 0;
	unsigned pos = 0;
	buf[pos++] = (uint8_t)(((version & 0xf) << 4) | (tn & 0x7));
	buf[pos++] = (uint8_t)(fn >> 24);                      /* osmo_store32be */
	buf[pos++] = (uint8_t)(fn >> 16);
	buf[pos++] = (uint8_t)(fn >> 8);
	buf[pos++] = (uint8_t)fn;
	buf[pos++] = rssi >= 255.0 ? 255 : (rssi > 0.0 ? (uint8_t)rssi : 0);
	int toa_int = orc_trxd_toa256(toa);
	buf[pos++] = (uint8_t)((unsigned)toa_int >> 8);        /* osmo_store16be */
	buf[pos++] = (uint8_t)toa_int;
	if (version == 1) {
		int16_t ci_cb = orc_trxd_ci_cb(ci);
		unsigned mod = modulation_8psk ? (0x4u | (tss & 0x1u)) : (0x0u | (tss & 0x3u));   /* TRXD_MODULATION_8PSK / _GMSK */
		buf[pos++] = (uint8_t)(((idle ? 1u : 0u) << 7) | ((mod & 0xf) << 3) | (tsc & 0x7));
		buf[pos++] = (uint8_t)((uint16_t)ci_cb >> 8);
		buf[pos++] = (uint8_t)ci_cb;
		if (!idle) {                                       /* :106-107 */
			orc_trxd_soft_u8(buf + pos, rx_burst, nbits);
			pos += nbits;
		}
		return (int)pos;
	}
	orc_trxd_soft_u8(buf + pos, rx_burst, nbits);
	pos += nbits;
	buf[pos++] = 0;                                        /* uninitialised in the reference */
	buf[pos++] = '\0';                                     /* :87 */
	return (int)pos;
}

/* ------------------------------------------------------------------------------------------
 * Channelizer.cpp / ChannelizerBase.cpp.  FFTW (arch/common/fft.c:55-114) is a third-party
 * dependency absent here ("fftw3f", unpinned via pkg-config, configure.ac:291); its M-point
 * forward DFT is restated directly:  X[k] = sum_j x[j] * exp(-2*pi*i*j*k/M).
 * For M = 4 the butterflies are exact +-1/+-j operations.
 * ---------------------------------------------------------------------------------------- */
struct orc_channelizer {
	int m, block_len, h_len;
	float *sub;      /* m x h_len real taps (reversed) */
	orc_cf *hist;    /* m x h_len */
	orc_cf *fir_in;  /* m x (h_len + block_len) */
	orc_cf *fir_out; /* m x block_len */
};

/* ChannelizerBase.cpp:37-43 */
static float ch_sinc(float x)
{
	if (x == 0.0f)
		return 0.999999999999f;
	return sin(M_PI * x) / (M_PI * x);
}

/* ChannelizerBase.cpp:68-134 initFilters */
orc_channelizer *orc_channelizer_new(int m, int block_len, int h_len)
{
	orc_channelizer *c = calloc(1, sizeof(*c));
	c->m = m; c->block_len = block_len; c->h_len = h_len;
	size_t proto_len = (size_t)m * h_len;
	float *proto = calloc(proto_len, sizeof(float));
	c->sub = calloc(proto_len, sizeof(float));
	c->hist = calloc((size_t)m * h_len, sizeof(orc_cf));
	c->fir_in = calloc((size_t)m * (h_len + block_len), sizeof(orc_cf));
	c->fir_out = calloc((size_t)m * block_len, sizeof(orc_cf));

	float sum = 0.0f, scale = 0.0f;
	float midpt = (float)(proto_len - 1.0) / 2.0;
	float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
	for (size_t i = 0; i < proto_len; i++) {
		proto[i] = ch_sinc(((float)i - midpt) / (float)m);
		proto[i] *= a0 -
			    a1 * cos(2 * M_PI * i / (proto_len - 1)) +
			    a2 * cos(4 * M_PI * i / (proto_len - 1)) -
			    a3 * cos(6 * M_PI * i / (proto_len - 1));
		sum += proto[i];
	}
	scale = (float)m / sum;
	for (int i = 0; i < h_len; i++)
		for (int n = 0; n < m; n++)
			c->sub[(size_t)n * h_len + (h_len - 1 - i)] = proto[(size_t)i * m + n] * scale;
	free(proto);
	return c;
}

void orc_channelizer_free(orc_channelizer *c)
{
	if (!c) return;
	free(c->sub); free(c->hist); free(c->fir_in); free(c->fir_out); free(c);
}

const float *orc_channelizer_subfilter(const orc_channelizer *c, int path)
{
	return &c->sub[(size_t)path * c->h_len];
}

/* Channelizer.cpp:74-99 */
int orc_channelizer_rotate(orc_channelizer *c, const orc_cf *in, int len, orc_cf *out)
{
	const int m = c->m, bl = c->block_len, hl = c->h_len;
	if (len != bl * m)
		return -1;

	/* deinterleave :37-48: path (m-1-n) takes samples n, n+m, ... */
	for (int i = 0; i < bl; i++)
		for (int n = 0; n < m; n++)
			c->fir_in[(size_t)(m - 1 - n) * (hl + bl) + hl + i] = in[(size_t)i * m + n];

	for (int p = 0; p < m; p++) {
		orc_cf *x = &c->fir_in[(size_t)p * (hl + bl) + hl];
		memcpy(x - hl, &c->hist[(size_t)p * hl], (size_t)hl * sizeof(orc_cf));        /* :87 */
		memcpy(&c->hist[(size_t)p * hl], x + bl - hl, (size_t)hl * sizeof(orc_cf));   /* :88 */
		/* convolve_real(start=0, len=blockLen): y[i] = sum_k x[i-(hl-1)+k]*h[k] */
		const float *h = &c->sub[(size_t)p * hl];
		for (int i = 0; i < bl; i++) {
			float yr = 0.0f, yi = 0.0f;
			for (int k = 0; k < hl; k++) {
				orc_cf xv = x[i - (hl - 1) + k];
				yr += xv.re * h[k];
				yi += xv.im * h[k];
			}
			c->fir_out[(size_t)p * bl + i] = cf(yr, yi);
		}
	}

	/* cxvec_fft :96 -- for each time sample an m-point forward DFT across paths */
	for (int t = 0; t < bl; t++) {
		if (m == 4) {
			orc_cf x0 = c->fir_out[0 * bl + t], x1 = c->fir_out[1 * bl + t];
			orc_cf x2 = c->fir_out[2 * bl + t], x3 = c->fir_out[3 * bl + t];
			orc_cf t1 = cf(x0.re + x2.re, x0.im + x2.im), t2 = cf(x0.re - x2.re, x0.im - x2.im);
			orc_cf t3 = cf(x1.re + x3.re, x1.im + x3.im), t4 = cf(x1.re - x3.re, x1.im - x3.im);
			out[0 * bl + t] = cf(t1.re + t3.re, t1.im + t3.im);
			out[2 * bl + t] = cf(t1.re - t3.re, t1.im - t3.im);
			/* X1 = t2 - j*t4, X3 = t2 + j*t4 */
			out[1 * bl + t] = cf(t2.re + t4.im, t2.im - t4.re);
			out[3 * bl + t] = cf(t2.re - t4.im, t2.im + t4.re);
		} else {
			for (int k = 0; k < m; k++) {
				double sr = 0.0, si = 0.0;
				for (int j = 0; j < m; j++) {
					double ang = -2.0 * M_PI * (double)((j * k) % m) / m;
					orc_cf x = c->fir_out[(size_t)j * bl + t];
					sr += x.re * cos(ang) - x.im * sin(ang);
					si += x.re * sin(ang) + x.im * cos(ang);
				}
				out[(size_t)k * bl + t] = cf((float)sr, (float)si);
			}
		}
	}
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Viterbi alternative of pullRadioVector (cfg->use_va): demodAnyBurst_va(), Transceiver.cpp:620-645,
 * on top of Transceiver52M/grgsm_vitac/{grgsm_vitac.cpp, viterbi_detector.cc} (gr-gsm's MLSE receiver,
 * 4 samples per symbol, 5-symbol channel, 16 states).  std::complex<float> arithmetic is spelled out:
 * operator* = (ar*br - ai*bi, ar*bi + ai*br); division by (len, 0) = component-wise division (libgcc
 * __divsc3 with a zero imaginary divisor); abs() = cabsf; std::pow(float, int) = pow in double.
 * ---------------------------------------------------------------------------------------- */
#define VA_OSR 4
#define VA_CIR 5                                  /* CHAN_IMP_RESP_LENGTH, constants.h */
#define VA_BURST 148                              /* BURST_SIZE */
#define VA_AB_BURST (8 + 41 + 36 + 3)             /* grgsm_vitac.cpp:104 */
#define VA_TRAIN_BEGINNING 5
#define VA_TRAIN_POS (3 + 58 + 5)                 /* TRAIN_POS */

static orc_cf va_norm_seq[8][26];                 /* d_norm_training_seq (TSC 0..7) */
static orc_cf va_acc_seq[41];                     /* d_acc_training_seq */
static int va_ready;

/* grgsm_vitac.cpp:122-145 gmsk_mapper, then conj (:57-79) */
static void va_gmsk_map(const char *bits, int n, orc_cf start, orc_cf *out)
{
	const orc_cf j = cf(0.0f, 1.0f);
	out[0] = start;
	int prev = 2 * (bits[0] - '0') - 1;
	for (int i = 1; i < n; i++) {
		int cur = 2 * (bits[i] - '0') - 1;
		int enc = cur * prev;
		out[i] = cmul(cmul(j, cf((float)enc, 0.0f)), out[i - 1]);
		prev = cur;
	}
	for (int i = 0; i < n; i++)
		out[i] = cf(out[i].re, -out[i].im);
}

static void va_init(void)
{
	if (va_ready) return;
	va_gmsk_map(RACH_BITS[0], 41, cf(0.0f, -1.0f), va_acc_seq);            /* ACCESS_BITS, constants.h:91-95 */
	for (int t = 0; t < 8; t++)
		va_gmsk_map(TSC_BITS[t], 26, TSC_BITS[t][0] == '0' ? cf(1.0f, 0.0f) : cf(-1.0f, 0.0f), va_norm_seq[t]);
	va_ready = 1;
}

/* grgsm_vitac.cpp:147-155 correlate_sequence; x = 0 outside [0, n) */
static orc_cf va_correlate(const orc_cf *seq, int len, const orc_cf *x, int n, int pos)
{
	orc_cf r = cf(0.0f, 0.0f);
	for (int ii = 0; ii < len; ii++) {
		int j = pos + ii * VA_OSR;
		orc_cf t = cmul(seq[ii], (j >= 0 && j < n) ? x[j] : cf(0.0f, 0.0f));
		r.re += t.re;
		r.im += t.im;
	}
	return cf(r.re / (float)len, -r.im / (float)len);
}

/* grgsm_vitac.cpp:183-232 get_chan_imp_resp: returns the first sample of the impulse response */
static int va_chan_imp_resp(const orc_cf *x, int n, orc_cf *cir, int start_pos, int stop_pos,
			    const orc_cf *tseq, int tseqlen)
{
	const int nw = stop_pos - start_pos, wl = VA_CIR * VA_OSR;
	orc_cf corr[256];
	float power[256], energy[256];
	for (int ii = 0; ii < nw; ii++) {
		corr[ii] = va_correlate(tseq, tseqlen, x, n, start_pos + ii);
		power[ii] = (float)pow((double)hypotf(corr[ii].re, corr[ii].im), 2.0);   /* abs(): cabsf = hypotf */
	}
	float ws = 0;
	int ne = 0;
	for (int i = 0; i < wl; i++)
		ws += power[i];
	energy[ne++] = ws;
	for (int i = wl; i < nw; i++) {
		ws += power[i] - power[i - wl];
		energy[ne++] = ws;
	}
	int best = 0;                                  /* std::max_element: first largest */
	for (int i = 1; i < ne; i++)
		if (energy[best] < energy[i]) best = i;
	for (int ii = 0; ii < wl; ii++)
		cir[ii] = corr[best + ii];
	return start_pos + best;
}

/* viterbi_detector.cc:62-392.  Regular structure of its 32 hand-written add-compare-select statements:
 * new state s comes from old states p = s>>1 and p+8; on imaginary steps candidate 1 is
 * old[p] +- in -+ inc[IA[p]], candidate 2 old[p+8] +- in +- inc[IB[p]] (upper signs for even s); on real steps
 * old[p] -+ in -+ inc[7-p], old[p+8] -+ in +- inc[p].  Evaluated left to right as written there. */
static const int VA_IA[8] = { 2, 3, 0, 1, 6, 7, 4, 5 }, VA_IB[8] = { 5, 4, 7, 6, 1, 0, 3, 2 };

void orc_va_viterbi(const orc_cf *in, unsigned n, const orc_cf *rhh, unsigned start_state,
		    const unsigned *stop_states, unsigned nstops, float *out)
{
	float inc[8], pm1[16], pm2[16], *oldm = pm1, *newm = pm2;
	float *trans = malloc((size_t)n * 16 * sizeof(float));
	int real_imag = 0;
	for (int i = 0; i < 16; i++) pm1[i] = (-10e30);
	if (start_state < 16) pm1[start_state] = 0;   /* the reference writes out of bounds for larger values */
	for (int m = 0; m < 8; m++) {
		float v = (m & 1) ? rhh[1].im : -rhh[1].im;
		v = (m & 2) ? v + rhh[2].re : v - rhh[2].re;
		v = (m & 4) ? v + rhh[3].im : v - rhh[3].im;
		inc[m] = v + rhh[4].re;
	}
	for (unsigned k = 0; k < n; k++) {
		real_imag = !(k & 1);                      /* even samples: imaginary part */
		float sym = real_imag ? in[k].im : in[k].re;
		for (int s = 0; s < 16; s++) {
			int p = s >> 1, odd = s & 1;
			float c1, c2;
			if (real_imag) {
				if (!odd) { c1 = oldm[p] + sym - inc[VA_IA[p]]; c2 = oldm[p + 8] + sym + inc[VA_IB[p]]; }
				else      { c1 = oldm[p] - sym + inc[VA_IA[p]]; c2 = oldm[p + 8] - sym - inc[VA_IB[p]]; }
			} else {
				if (!odd) { c1 = oldm[p] - sym - inc[7 - p]; c2 = oldm[p + 8] - sym + inc[p]; }
				else      { c1 = oldm[p] + sym + inc[7 - p]; c2 = oldm[p + 8] + sym - inc[p]; }
			}
			float d = c2 - c1;
			newm[s] = (d < 0) ? c1 : c2;
			trans[k * 16 + s] = d;
		}
		float *t = oldm; oldm = newm; newm = t;
	}
	unsigned best = stop_states[0];
	float bm = oldm[best];
	for (unsigned i = 1; i < nstops; i++)
		if (oldm[stop_states[i]] > bm) { bm = oldm[stop_states[i]]; best = stop_states[i]; }
	unsigned state = best;
	int out_bit = 0;
	for (unsigned k = n; k-- > 0;) {
		float tv = trans[k * 16 + state];
		int decision = tv > 0;
		out[k] = (decision != out_bit) ? -tv : tv;
		int parity = ((state >> 1) ^ state) & 1;   /* parity_table: 0 1 1 0 0 1 1 0 ... */
		out_bit = out_bit ^ real_imag ^ parity;
		state = (state >> 1) + (decision ? 8 : 0); /* prev_table */
		real_imag = !real_imag;
	}
	free(trans);
}

/* grgsm_vitac.cpp:82-108 detect_burst_generic: x = 0 outside [0, n) */
static void va_detect_burst(const orc_cf *x, int n, const orc_cf *cir, int burst_start, int burst_size,
			    unsigned start_state, float *out)
{
	const int fl = VA_CIR * VA_OSR;
	orc_cf rt[VA_CIR * VA_OSR], rhh[VA_CIR], *filt = malloc((size_t)burst_size * sizeof(orc_cf));
	for (int k = fl - 1; k >= 0; k--) {            /* autocorrelation :158-166 */
		rt[k] = cf(0.0f, 0.0f);
		for (int i = k; i < fl; i++) {
			orc_cf t = cmul(cir[i], cf(cir[i - k].re, -cir[i - k].im));
			rt[k].re += t.re;
			rt[k].im += t.im;
		}
	}
	for (int ii = 0; ii < VA_CIR; ii++)
		rhh[ii] = cf(rt[ii * VA_OSR].re, -rt[ii * VA_OSR].im);
	for (int m = 0; m < burst_size; m++) {         /* mafi :168-181 */
		int a = m * VA_OSR;
		filt[m] = cf(0.0f, 0.0f);
		for (int ii = 0; ii < fl; ii++) {
			if (a + ii >= burst_size * VA_OSR) break;
			int j = burst_start + a + ii;
			orc_cf t = cmul((j >= 0 && j < n) ? x[j] : cf(0.0f, 0.0f), cir[ii]);
			filt[m].re += t.re;
			filt[m].im += t.im;
		}
	}
	const unsigned stops[2] = { 4, 12 };
	orc_va_viterbi(filt, (unsigned)burst_size, rhh, start_state, stops, 2, out);
	free(filt);
}

/* Transceiver.cpp:782-784 + :620-645: scaleVector(burst, scale) then demodAnyBurst_va().  soft: 156 values,
 * +-127 for the demodulated bits (148 normal / 88 access), 0 for the rest (the reference leaves bits 88..147 of an
 * access burst uninitialised).  Samples outside the burst read as 0 (the reference reads up to 2 samples past its
 * 625-sample vector for the latest burst position).  Returns the burst start (samples), -1 on a bad argument. */
int orc_demod_any_burst_va(const orc_cf *burst, int n, int type, int tsc, int max_toa, float scale, float *soft)
{
	va_init();
	if (tsc < 0 || tsc > 7) return -1;
	orc_cf *x = malloc((size_t)n * sizeof(orc_cf)), cir[VA_CIR * VA_OSR];
	for (int i = 0; i < n; i++)
		x[i] = cmul(burst[i], cf(scale, 0.0f));    /* scaleVector :1198-1205 */
	float out[VA_BURST];
	int nb, start;
	if (type == ORC_TSC) {
		const int center = VA_TRAIN_POS;           /* get_norm_chan_imp_resp :263-272 */
		start = va_chan_imp_resp(x, n, cir, (center - 5) * VA_OSR + 1, (center + 5 + VA_CIR) * VA_OSR,
					 &va_norm_seq[tsc][VA_TRAIN_BEGINNING], 26 - 2 * VA_TRAIN_BEGINNING) - center * VA_OSR;
		if (start < 0) start = 0;
		nb = VA_BURST;
		va_detect_burst(x, n, cir, start, nb, 3, out);
	} else {
		const int center = 8 + 5;                  /* get_access_imp_resp(..., max_delay = 0) :244-253 */
		start = va_chan_imp_resp(x, n, cir, (center - 5) * VA_OSR + 1, (center + 5 + VA_CIR + 0) * VA_OSR,
					 &va_acc_seq[VA_TRAIN_BEGINNING], 41 - 2 * VA_TRAIN_BEGINNING) - center * VA_OSR;
		if (start < 0) start = 0;
		nb = VA_AB_BURST;
		va_detect_burst(x, n, cir, start, nb, (unsigned)max_toa, out);   /* rach_max_toa lands in start_state (:633) */
	}
	for (int i = 0; i < 156; i++)
		soft[i] = 0.0f;
	for (int i = 0; i < nb; i++)
		soft[i] = (float)((out[i] > 0 ? -127 : 127) * -1);   /* sbit "pre flip" (:107) and "* -1" (:638) */
	free(x);
	return start;
}

/* ------------------------------------------------------------------------------------------
 * Transceiver::pullRadioVector() around its DSP calls, Transceiver.cpp:665-815
 * ---------------------------------------------------------------------------------------- */
void orc_rx_state_init(orc_rx_state *st)
{
	memset(st, 0, sizeof(*st));                      /* Transceiver.cpp:64-68; std::vector<float>(20): zeros */
}

/* avgVector::insert, radioVector.cpp:97-108 */
static void rx_noise_insert(orc_rx_state *st, float val)
{
	if (st->itr >= ORC_NOISE_CNT)
		st->itr = 0;
	st->noises[st->itr++] = val;
}

/* avgVector::avg, radioVector.cpp:84-95 */
static float rx_noise_avg(const orc_rx_state *st)
{
	float val = 0.0;
	for (size_t i = 0; i < ORC_NOISE_CNT; i++)
		val += st->noises[i];
	return val / (float)ORC_NOISE_CNT;
}

int orc_pull_radio_vector(orc_rx_state *st, int type, uint32_t fn, uint8_t tn, float pow_avg, int rc, const orc_ebp *ebp,
			  const float *soft, int nsoft, double full_scale, double rssi_offset, orc_ul_burst_ind *bi)
{
	float avg;
	/* :693-704 */
	bi->nbits = 0;
	bi->fn = fn;
	bi->tn = tn;
	bi->rssi = 0.0;
	bi->toa = 0.0;
	bi->noise = 0.0;
	bi->idle = 0;
	bi->modulation = 0;
	bi->tss = 0;
	bi->tsc = 0;
	bi->ci = 0.0;
	if (type == ORC_OFF)                             /* :713-717 */
		return -2;                               /* -ENOENT */
	if (st->muted)                                   /* :719-721 */
		goto ret_idle;
	avg = sqrtf(pow_avg);                            /* :741 (float sqrt of a float: the C++ overload) */
	if (type == ORC_IDLE) {                          /* :743-748 */
		rx_noise_insert(st, avg);
		st->noise_lev = rx_noise_avg(st);
	}
	bi->rssi = 20.0 * log10(full_scale / avg) + rssi_offset;              /* :751 */
	bi->noise = 20.0 * log10(full_scale / st->noise_lev) + rssi_offset;   /* :752 */
	if (type == ORC_IDLE)                            /* :754-755 */
		goto ret_idle;
	if (rc <= 0) {                                   /* :769-781 */
		if (rc == -ORC_SIGERR_CLIP)
			st->rx_clipping++;
		else if (rc != ORC_SIGERR_NONE)
			st->rx_no_burst_detected++;
		goto ret_idle;
	}
	bi->toa = ebp->toa;                              /* :789-791 */
	bi->tsc = ebp->tsc;
	bi->ci = ebp->ci;
	if (nsoft == 444) {                              /* :794-800 */
		bi->modulation = 1;
		bi->nbits = 444;
	} else {
		bi->modulation = 0;
		bi->nbits = 148;                         /* gSlotLen */
	}
	orc_vector_slicer(bi->rx_burst, soft, bi->nbits);                     /* :803 */
	return 0;
ret_idle:
	bi->idle = 1;                                    /* :808 */
	return 0;
}
