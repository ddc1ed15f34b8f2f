// trx_tables.cpp -- host-side generation of the device table blob (product code, not the oracle).
//
// Builds what sigProcLibSetup() builds (Transceiver52M/sigProcLib.cpp:2139-2172): it runs once at
// init on the host, exactly like the reference, and the result is uploaded / RCCL-broadcast to
// the GPUs.  Each generator cites the reference lines whose arithmetic (operand order and
// float/double promotion points) it keeps, so that the tables are bit-identical to the
// reference's and every decision the kernels take on them matches.
#include <cmath>
#include <cstring>
#include <vector>

#include "trx_tables.h"

namespace {

const float kPiF = (float)M_PI;                    // sigProcLib.cpp:55

// Complex<float> arithmetic as the reference defines it (Transceiver52M/Complex.h:70-150)
struct cx {
	float r = 0.0f, i = 0.0f;
	cx() = default;
	cx(float re, float im) : r(re), i(im) {}
	cx operator*(const cx &a) const { return cx(r * a.r - i * a.i, r * a.i + i * a.r); }
	cx operator*(float a) const { return cx(r * a, i * a); }
	cx conj() const { return cx(r, -i); }
	float norm2() const { return i * i + r * r; }
	float abs() const { return std::sqrt(norm2()); }
	cx inv() const { float n = norm2(); return cx(r / n, -i / n); }
	cx operator/(const cx &a) const { return *this * a.inv(); }
};
typedef std::vector<cx> cvec;

// 3GPP TS 45.002 training / synchronisation sequences (GSM/GSMCommon.cpp:35-68)
const char *const kTsc[8] = {
	"00100101110000100010010111", "00101101110111100010110111", "01000011101110100100001110",
	"01000111101101000100011110", "00011010111001000001101011", "01001110101100000100111010",
	"10100111110110001010011111", "11101111000100101110111100" };
const char *const kEdgeTsc[8] = {
	"111111001111111001111001001001111111111111001111111111001111111001111001001001",
	"111111001111001001111001001001111001001001001111111111001111001001111001001001",
	"111001111111111111001001001111001001001111001111111001111111111111001001001111",
	"111001111111111001001001001111001001111001111111111001111111111001001001001111",
	"111111111001001111001111001001001111111001111111111111111001001111001111001001",
	"111001111111001001001111001111001001111111111111111001111111001001001111001111",
	"001111001111111001001001001001111001001111111111001111001111111001001001001001",
	"001001001111001001001001111111111001111111001111001001001111001001001001111111" };
const char *const kDummyTsc = "01110001011100010111000101";
const char *const kRach[3] = {
	"01001011011111111001100110101010001111000",
	"01010100111110001000011000101111001001101",
	"11101111001001110101011000001101101110111" };
const char *const kSch = "1011100101100010000001000000111100101101010001010111011000011011";

std::vector<int> bits_of(const char *s)
{
	std::vector<int> b;
	for (; *s; ++s) b.push_back(*s == '1');
	return b;
}

class TableBuilder {
public:
	explicit TableBuilder(trx_tables *t) : t_(t) {}

	void build()
	{
		std::memset(t_, 0, sizeof(*t_));
		t_->magic = TRX_TABLES_MAGIC;
		t_->version = TRX_TABLES_VERSION;
		t_->fused_u0 = TRX_FUSED_U0;
		t_->fused_nt = TRX_FUSED_NT;
		sinc_table();
		rotation_tables();
		pulse_1sps();
		for (int i = 0; i < 3; i++)
			sync_sequence(&t_->seq[TRX_SEQ_RACH0 + i], kRach[i], 40, 20.5);   // :2147-2149
		sync_sequence(&t_->seq[TRX_SEQ_SCH], kSch, 64, 32.5);                      // :2151
		midamble(&t_->seq[TRX_SEQ_DUMMY], kDummyTsc);                              // :2152
		for (int tsc = 0; tsc < 8; tsc++) {                                        // :2154-2157
			midamble(&t_->seq[TRX_SEQ_TSC0 + tsc], kTsc[tsc]);
			edge_midamble(&t_->seq[TRX_SEQ_EDGE0 + tsc], kEdgeTsc[tsc]);
		}
		delay_filters();                                                           // :2159
		polyphase(1, 4, 16, 1.0f, &t_->dec_taps[0], 16);                           // :2161-2162
		polyphase(65, 48, 16, 1.0f, &t_->rs6548_taps[0][0], 16);                   // radioInterfaceMulti.cpp:35-36
		channelizer_filters(4, 16);                                                // radioInterfaceMulti.cpp:42
		static const double inv[5] = { 0.15884, -0.43176, 1.00000, -0.42608, 0.14882 };   // :414-419
		for (int i = 0; i < 5; i++) t_->c0_inv[i] = (float)inv[i];
		sincv_table();
		composite_filters();
		edge_lo_filters();
		edge_constants();
		unit_structure();
	}

	// Which sequences are "+-1 rotated by k*pi/2 with an fp64 phase residue" tap by tap (the premise of corr_unit(),
	// trx_device.h): even taps (+-1, e), odd taps (e, +-1), |e| <= 1e-13 (2^17 * 1e-13 = 1.3e-8 stays below a quarter ulp, 2^-26:
	// the exactness argument of corr_unit(); the 64-tap SCH sequence's residue reaches 7e-14 at its last taps, the others 4.6e-14).
	void unit_structure()
	{
		t_->unit_ok = 0;
		for (int s = 0; s < TRX_NSEQ; s++) {
			const trx_seq &q = t_->seq[s];
			bool ok = q.n > 0 && q.n <= 64;
			uint64_t neg = 0;
			for (int k = 0; ok && k < q.n; k++) {
				const float one = (k & 1) ? q.taps[k].im : q.taps[k].re;
				const float eps = (k & 1) ? q.taps[k].re : q.taps[k].im;
				ok = (one == 1.0f || one == -1.0f) && std::fabs(eps) <= 1e-13f;
				if (one < 0.0f)
					neg |= 1ull << k;
			}
			t_->unit_neg[s] = ok ? neg : 0;
			if (ok)
				t_->unit_ok |= 1u << s;
		}
	}

private:
	trx_tables *t_;
	float sinc_[1025];
	cx rot1_[157];
	float c0_1sps_[4];

	// generateSincTable, sigProcLib.cpp:981-988
	void sinc_table()
	{
		for (int i = 0; i < 1024; i++) {
			double x = (double)i / 1024 * 8 * M_PI;
			double y = std::sin(x) / x;
			sinc_[i] = std::isnan(y) ? 1.0 : y;
		}
		sinc_[1024] = 0.0f;
	}

	// sinc(), sigProcLib.cpp:990-998: the quotient is a double, floorf() sees it rounded to float
	float sinc(float x) const
	{
		if (std::fabs(x) >= 8 * M_PI)
			return 0.0f;
		int index = (int)floorf(std::fabs(x) / (8 * M_PI) * 1024);
		return sinc_[index];
	}

	// initGMSKRotationTables, sigProcLib.cpp:206-215 (1 SPS tables only: all sequences and the
	// derotation run at 1 SPS, :2147-2156, :2066)
	void rotation_tables()
	{
		double phase = 0.0;
		for (int i = 0; i < 157; i++) {
			rot1_[i] = cx(std::cos(phase), std::sin(phase));
			t_->rrot1[i].re = std::cos(-phase);
			t_->rrot1[i].im = std::sin(-phase);
			phase += M_PI / 2.0;
		}
	}

	// generateGSMPulse(1), sigProcLib.cpp:519-533
	void pulse_1sps()
	{
		const int len = 4, sps = 1;
		float center = (float)(len - 1.0) / 2.0;
		float energy = 0.0f;
		for (int i = 0; i < len; i++) {
			float arg = ((float)i - center) / (float)sps;
			c0_1sps_[i] = 0.96 * std::exp(-1.1380 * arg * arg - 0.527 * arg * arg * arg * arg);
			energy += cx(c0_1sps_[i], 0.0f).norm2();
		}
		float avg = sqrtf(energy / sps);
		for (int i = 0; i < len; i++)
			c0_1sps_[i] /= avg;
	}

	// modulateBurst(bits, 0, 1, emptyPulse=true) -> rotateBurst, sigProcLib.cpp:558-580
	cvec rotated_symbols(const std::vector<int> &bits, size_t first, size_t count) const
	{
		cvec out(count);
		for (size_t i = 0; i < count; i++) {
			cx sym((float)(2.0 * (bits[first + i] & 1) - 1.0), 0.0f);
			cx v = rot1_[i] * sym;                 // GMSKRotate, complex branch :253-257
			out[i] = cx(0.0f + v.r * 1.0f, 0.0f + v.i * 1.0f);   // 1-tap "empty" pulse, mac_real
		}
		return out;
	}

	// modulateBurst(bits, 0, 1, false) -> modulateBurstBasic, sigProcLib.cpp:938-967
	cvec shaped_symbols(const std::vector<int> &bits) const
	{
		const size_t n = bits.size();
		cvec sym(n), out(n);
		for (size_t i = 0; i < n; i++)
			sym[i] = rot1_[i] * (float)(2.0 * (bits[i] & 1) - 1.0);   // real branch :247-251
		for (size_t i = 0; i < n; i++) {               // START_ONLY: y[i] = sum_k x[i-3+k]*c0[k]
			cx acc;
			for (int k = 0; k < 4; k++) {
				long j = (long)i - 3 + k;
				cx x = (j >= 0) ? sym[j] : cx();
				acc.r += x.r * c0_1sps_[k];
				acc.i += x.i * c0_1sps_[k];
			}
			out[i] = acc;
		}
		return out;
	}

	// convolve(x, h, NULL, NO_DELAY) with complex taps, sigProcLib.cpp:318-323 + convolve_base.c:72-85
	static cvec correlate_same(const cvec &x, const cvec &h)
	{
		const long n = x.size(), H = h.size(), start = H / 2;
		cvec y(n);
		for (long i = 0; i < n; i++) {
			cx acc;
			for (long k = 0; k < H; k++) {
				long j = i + start - (H - 1) + k;
				cx xv = (j >= 0 && j < n) ? x[j] : cx();
				acc.r += xv.r * h[k].r - xv.i * h[k].i;
				acc.i += xv.r * h[k].i + xv.i * h[k].r;
			}
			y[i] = acc;
		}
		return y;
	}

	// interpolatePoint, sigProcLib.cpp:1100-1118
	cx interpolate(const cvec &s, float ix) const
	{
		int start = (int)(floorf(ix) - 10);
		if (start < 0) start = 0;
		int end = (int)(floorf(ix) + 11);
		if ((unsigned)end > s.size() - 1) end = s.size() - 1;
		cx p;
		for (int i = start; i < end; i++) {
			cx v = s[i] * sinc(kPiF * (i - ix));
			p.r += v.r;
			p.i += v.i;
		}
		return p;
	}

	// peakDetect, sigProcLib.cpp:1141-1186
	cx peak(const cvec &s, float *where) const
	{
		float best = 0.0f, at = -1;
		for (size_t i = 0; i < s.size(); i++) {
			float p = s[i].norm2();
			if (p > best) { best = p; at = i; }
		}
		float early = at - 1, late = at + 1, incr = 0.5;
		while (incr > 1.0 / 1024.0) {
			float pe = interpolate(s, early).norm2();
			float pl = interpolate(s, late).norm2();
			if (pe < pl) early += incr;
			else if (pe > pl) early -= incr;
			else break;
			incr /= 2.0;
			late = early + 2.0;
		}
		at = early + 1.0;
		*where = at;
		return interpolate(s, at);
	}

	void store(trx_seq *dst, const cvec &taps, const cvec &shaped, double toa_ref)
	{
		std::memset(dst, 0, sizeof(*dst));
		dst->n = taps.size();
		for (size_t i = 0; i < taps.size(); i++) {
			dst->taps[i].re = taps[i].r;
			dst->taps[i].im = taps[i].i;
		}
		float toa;
		cx gain = peak(correlate_same(shaped, taps), &toa);
		set_gain(dst, gain);
		dst->toa = toa - toa_ref;
	}

	static void set_gain(trx_seq *dst, cx gain)
	{
		dst->gain.re = gain.r;
		dst->gain.im = gain.i;
		cx gi = gain.inv();
		dst->gain_inv.re = gi.r;
		dst->gain_inv.im = gi.i;
		dst->ci_den = (dst->n - 1) * gain.abs();          // sigProcLib.cpp:1629
		dst->ci_den_inv = 1.0f / dst->ci_den;
	}

	// generateMidamble / generateDummyMidamble (sps = 1), sigProcLib.cpp:1227-1299, :1301-1370
	void midamble(trx_seq *dst, const char *bitstr)
	{
		std::vector<int> bits = bits_of(bitstr);
		cvec mid = rotated_symbols(bits, 5, 16);            // segment(5,16)
		cvec shaped = shaped_symbols(bits);
		for (auto &v : mid) v = v * cx(-1.0f, 0.0f);         // :1257
		for (auto &v : shaped) v = v * cx(0.0f, 1.0f);       // :1258
		for (auto &v : mid) v = v.conj();                    // :1260
		store(dst, mid, shaped, 13.5);
	}

	// generateRACHSequence / generateSCHSequence (sps = 1), sigProcLib.cpp:1405-1465, :1467-1527
	void sync_sequence(trx_seq *dst, const char *bitstr, size_t corr_bits, double toa_ref)
	{
		std::vector<int> bits = bits_of(bitstr);
		cvec shaped = shaped_symbols(bits);
		cvec taps = rotated_symbols(bits, 0, corr_bits);
		for (auto &v : taps) v = v.conj();
		store(dst, taps, shaped, toa_ref);
	}

	// generateEdgeMidamble, sigProcLib.cpp:1372-1403 (mapEdgeSymbols :713-729, rotateEdgeBurst :672-689)
	void edge_midamble(trx_seq *dst, const char *bitstr)
	{
		static const double psk8[8][2] = {
			{ -0.70710678, 0.70710678 }, { 0.0, -1.0 }, { 0.0, 1.0 }, { 0.70710678, -0.70710678 },
			{ -1.0, 0.0 }, { -0.70710678, -0.70710678 }, { 0.70710678, 0.70710678 }, { 1.0, 0.0 } };
		std::vector<int> bits = bits_of(bitstr);
		std::memset(dst, 0, sizeof(*dst));
		dst->n = 16;
		for (size_t i = 0; i < 16; i++) {
			const int *b = &bits[15 + 3 * i];
			unsigned idx = (b[0] & 1) | ((b[1] & 1) << 1) | ((b[2] & 1) << 2);
			cx sym((float)psk8[idx][0], (float)psk8[idx][1]);
			float phase = i * 3.0f * M_PI / 8.0f;
			cx v = (sym * cx(cosf(phase), sinf(phase))).conj();
			dst->taps[i].re = v.r;
			dst->taps[i].im = v.i;
		}
		const float k = 1.18;
		set_gain(dst, cx((float)-19.6432 / k, (float)19.5006 / k));           // :1397
		dst->toa = 0;
	}

	// generateDelayFilters, sigProcLib.cpp:1005-1044
	void delay_filters()
	{
		const int h_len = TRX_DELAY_HLEN;
		const float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
		for (int i = 0; i < TRX_DELAY_FILTS; i++) {
			float *h = t_->delay_filt[i];
			float sum = 0.0f;
			for (int n = 0; n < h_len; n++) {
				float k = (float)n;
				float tap = sinc(kPiF * (k - (float)h_len / 2.0 - (float)i / TRX_DELAY_FILTS));
				float win = a0 - a1 * std::cos(2 * M_PI * n / (h_len - 1)) +
					    a2 * std::cos(4 * M_PI * n / (h_len - 1)) -
					    a3 * std::cos(6 * M_PI * n / (h_len - 1));
				tap *= win;
				h[h_len - 1 - n] = tap;
				sum += tap;
			}
			for (int n = 0; n < h_len; n++)
				h[n] /= sum;
		}
	}

public:
	// Resampler::initFilters, Resampler.cpp:47-96 (+ its sinc :39-45)
	static void polyphase(size_t p, size_t q, size_t filt_len, float bw, float *out, size_t out_stride)
	{
		std::vector<float> proto(p * filt_len);
		const float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
		float cutoff = (p > q) ? (float)p : (float)q;
		float sum = 0.0f;
		float midpt = (proto.size() - 1) / 2.0;
		for (size_t i = 0; i < proto.size(); i++) {
			float x = ((float)i - midpt) / cutoff * bw;
			proto[i] = (x == 0.0) ? 0.9999999999 : std::sin(M_PI * x) / (M_PI * x);
			proto[i] *= a0 - a1 * std::cos(2 * M_PI * i / (proto.size() - 1)) +
				    a2 * std::cos(4 * M_PI * i / (proto.size() - 1)) -
				    a3 * std::cos(6 * M_PI * i / (proto.size() - 1));
			sum += proto[i];
		}
		float scale = p / sum;
		for (size_t i = 0; i < filt_len; i++)
			for (size_t n = 0; n < p; n++)
				out[n * out_stride + (filt_len - 1 - i)] = proto[i * p + n] * scale;   // stored reversed
	}

private:
	// ChannelizerBase::initFilters, ChannelizerBase.cpp:68-134 (+ its sinc :37-43)
	void channelizer_filters(size_t m, size_t h_len)
	{
		const size_t proto_len = m * h_len;
		std::vector<float> proto(proto_len);
		const float a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
		float sum = 0.0f;
		float midpt = (float)(proto_len - 1.0) / 2.0;
		for (size_t i = 0; i < proto_len; i++) {
			float x = ((float)i - midpt) / (float)m;
			proto[i] = (x == 0.0f) ? 0.999999999999f : (float)(std::sin(M_PI * x) / (M_PI * x));
			proto[i] *= a0 - a1 * std::cos(2 * M_PI * i / (proto_len - 1)) +
				    a2 * std::cos(4 * M_PI * i / (proto_len - 1)) -
				    a3 * std::cos(6 * M_PI * i / (proto_len - 1));
			sum += proto[i];
		}
		float scale = (float)m / sum;
		for (size_t i = 0; i < h_len; i++)
			for (size_t n = 0; n < m; n++)
				t_->chan_taps[n][h_len - 1 - i] = proto[i * m + n] * scale;
	}

	// Derived table for the fused demodulator: delayVector's 20-tap filter followed by the 16-tap /4 decimator
	// (sigProcLib.cpp:1060 then :1587-1601) is one 35-tap filter per delay index.  Products in double, rounded once.
	void composite_filters()
	{
		for (int f = 0; f <= TRX_DELAY_FILTS; f++) {
			for (int u = 0; u < 36; u++) {
				double acc = 0.0;
				for (int t = 0; t < 16; t++) {
					const int k = u - t;
					if (k < 0 || k >= TRX_DELAY_HLEN)
						continue;
					const double h = (f < TRX_DELAY_FILTS) ? (double)t_->delay_filt[f][k] : (k == 9 ? 1.0 : 0.0);
					acc += (double)t_->dec_taps[t] * h;
				}
				t_->comp_filt[f][u] = (u < 35) ? (float)acc : 0.0f;
			}
		}
	}

	// composite filters of the low-side partial outputs (decimator truncated to taps t >= t0), as composite_filters()
	void edge_lo_filters()
	{
		for (int f = 0; f <= TRX_DELAY_FILTS; f++)
			for (int t0 = 1; t0 <= 15; t0++)
				for (int u = 0; u < 36; u++) {
					double acc = 0.0;
					for (int t = t0; t < 16; t++) {
						const int k = u - t;
						if (k < 0 || k >= TRX_DELAY_HLEN)
							continue;
						const double h = (f < TRX_DELAY_FILTS) ? (double)t_->delay_filt[f][k] : (k == 9 ? 1.0 : 0.0);
						acc += (double)t_->dec_taps[t] * h;
					}
					t_->edge_lo[f][t0 - 1][u] = (u < 35) ? (float)acc : 0.0f;
				}
		// the four rows of the usual geometry, re-packed for the main filter loop (trx_tables.h)
		for (int f = 0; f <= TRX_DELAY_FILTS; f++)
			for (int i = 0; i < 4; i++)
				for (int k = 0; k < TRX_FUSED_NTP; k++) {
					// main part: taps u = U0 + k; early part: the lane's window starts 8 samples early, so its tap k is
					// u = k - (8 - U0), kept for u < U0
					const int ue = k - (8 - TRX_FUSED_U0);
					t_->edge8[f][i][k] = (k < TRX_FUSED_NT) ? t_->edge_lo[f][14 - 4 * i][TRX_FUSED_U0 + k] : 0.0f;
					t_->edge8[f][4 + i][k] = (ue >= 0 && ue < TRX_FUSED_U0) ? t_->edge_lo[f][14 - 4 * i][ue] : 0.0f;
				}
		// high side: the decimator truncated to taps t <= tm
		for (int f = 0; f <= TRX_DELAY_FILTS; f++)
			for (int tm = 0; tm <= 14; tm++)
				for (int u = 0; u < 36; u++) {
					double acc = 0.0;
					for (int t = 0; t <= tm; t++) {
						const int k = u - t;
						if (k < 0 || k >= TRX_DELAY_HLEN)
							continue;
						const double h = (f < TRX_DELAY_FILTS) ? (double)t_->delay_filt[f][k] : (k == 9 ? 1.0 : 0.0);
						acc += (double)t_->dec_taps[t] * h;
					}
					t_->edge_hi[f][tm][u] = (u < 35) ? (float)acc : 0.0f;
				}
	}

	// EDGE 8-PSK demodulator constants, with the float/double steps of the reference
	void edge_constants()
	{
		for (int i = 0; i < 16; i++) {                        // derotateEdgeBurst, sigProcLib.cpp:702-706
			float phase = (float)(i % 16) * 3.0f * M_PI / 8.0f;
			t_->edge_derot[i].re = cosf(phase);
			t_->edge_derot[i].im = -sinf(phase);
		}
		const float step = 2.0f * kPiF / 8.0f;                // computeEdgeCI, :2077
		t_->edge_step = step;
		for (int k = -4; k <= 4; k++) {                       // :2082-2083  complex(cos(phase), sin(phase)), float phase
			float phase = step * (float)k;
			t_->edge_ideal[k + 4].re = std::cos(phase);
			t_->edge_ideal[k + 4].im = std::sin(phase);
		}
		const double ph[2] = { -M_PI / 8.0, -M_PI / 4.0 };    // rotateBurst2, :582-588 (double phase)
		for (int j = 0; j < 2; j++) {
			t_->edge_rot2[j].re = std::cos(ph[j]);
			t_->edge_rot2[j].im = std::sin(ph[j]);
		}
	}

	// Every sinc() value interpolatePoint() can ask for: positions are multiples of 1/512, so
	// (i - ix) = +-q/512 exactly in float and sinc(M_PI_F * q/512) depends on q only.
	void sincv_table()
	{
		for (int q = 0; q < TRX_SINCV_LEN; q++) {
			float d = (float)q / 512.0f;
			t_->sincv[trx_sincv_swz(q)] = sinc(kPiF * d);
		}
	}
};

}  // namespace

void trx_polyphase_taps(unsigned p, unsigned q, unsigned filt_len, float bw, float *out)
{
	TableBuilder::polyphase(p, q, filt_len, bw, out, filt_len);
}

int trx_tables_generate(trx_tables *out)
{
	if (!out)
		return -1;
	TableBuilder(out).build();
	return 0;
}
