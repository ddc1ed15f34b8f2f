// sigProcLib.cpp -- host shim: the reference's sigProcLib receive-side API (sigProcLib.h) on top of the
// C ABI of include/trxhip.h.  Every computation happens on the MI355X; this file only moves buffers
// (hipMemcpyAsync on a per-thread stream) and converts between the reference's containers and the
// C-ABI records.  Errors map to the reference's conventions: detectAnyBurst() -> -SIGERR_INTERNAL,
// demodAnyBurst() -> NULL, pullRadioVectorBatch() -> -EIO (Transceiver.cpp:686).
//
// Two builds of this one source (osmo_trx_amd/build.py):
//   libtrxsigproc.so     -I<osmo-trx>/Transceiver52M -I<osmo-trx>/CommonLibs: "sigProcLib.h" is the reference's own
//                        header, so signalVector / SoftVector / estim_burst_params have the reference's layout and the
//                        exported symbols are the ones reference-compiled callers (Transceiver.o) import.  Only public
//                        methods of those classes are used (begin/end/size/bytes/isReal/clone, new signalVector(n),
//                        new SoftVector(n)); their out-of-line members come from the caller's own signalVector.o.
//   libtrxsigproc_sa.so  -Ihost/compat: the same API on look-alike containers in the inline namespace trxhip_sa, for
//                        boxes without an osmo-trx checkout (different mangled names: cannot be mixed up).
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "shim_internal.h"

TRX_SHIM_NS_BEGIN
namespace {

trxhip_ctx *g_ctx = nullptr;
std::mutex g_mu;

// Per-thread device scratch: the reference calls detect/demod concurrently from one RxUpper thread per
// ARFCN (Transceiver.cpp:308-314), so each thread gets its own stream and buffers.
struct Scratch {
	hipStream_t stream = nullptr;
	void *d_iq = nullptr;
	size_t iq_bytes = 0;
	trxhip_burst_params *d_prm = nullptr;
	trxhip_burst_result *d_res = nullptr;
	float *d_soft = nullptr;
	float *d_ebp = nullptr;
	size_t cap = 0;                  /* bursts */
	size_t soft_stride = 0;
	/* detect -> demod hand-over for the batch-of-1 calls: the fused kernel already produced the soft bits */
	const void *last_burst = nullptr;
	size_t last_size = 0;
	int last_sps = 0, last_rc = 0;
	estim_burst_params last_ebp;
	std::vector<float> last_soft;
	trxhip_burst_result last_res;

	static void drop(void *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
	template <class T> static void drop(T *&p) { void *q = p; drop(q); p = nullptr; }
	bool ensure(size_t n, size_t iq_b, size_t stride)
	{
		if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess)
			return false;
		if (iq_b > iq_bytes) {
			drop(d_iq);
			iq_bytes = 0;
			if (hipMalloc(&d_iq, iq_b) != hipSuccess) { d_iq = nullptr; return false; }
			iq_bytes = iq_b;
		}
		if (n > cap || stride > soft_stride) {
			const size_t nn = n > cap ? n : cap, ss = stride > soft_stride ? stride : soft_stride;
			/* nothing stale survives a failed re-allocation: pointers nulled, capacities zeroed */
			drop(d_prm); drop(d_res); drop(d_soft); drop(d_ebp);
			cap = 0;
			soft_stride = 0;
			if (hipMalloc((void **)&d_prm, nn * sizeof(trxhip_burst_params)) != hipSuccess ||
			    hipMalloc((void **)&d_res, nn * sizeof(trxhip_burst_result)) != hipSuccess ||
			    hipMalloc((void **)&d_soft, nn * ss * sizeof(float)) != hipSuccess ||
			    hipMalloc((void **)&d_ebp, nn * 4 * sizeof(float)) != hipSuccess) {
				drop(d_prm); drop(d_res); drop(d_soft); drop(d_ebp);
				return false;
			}
			cap = nn;
			soft_stride = ss;
		}
		return true;
	}
	/* host-fed pipeline of the batched calls: created on first use, re-created when the geometry changes */
	trxhip_hostpipe *pipe = nullptr;
	trxhip_hostpipe_cfg pipe_cfg;
	~Scratch()
	{
		if (pipe) trxhip_hostpipe_destroy(pipe);
		drop(d_iq); drop(d_prm); drop(d_res); drop(d_soft); drop(d_ebp);
		if (stream) (void)hipStreamDestroy(stream);
	}
};
thread_local Scratch tls;

bool h2d(void *d, const void *h, size_t n, hipStream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess; }
bool d2h(void *h, const void *d, size_t n, hipStream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess; }

size_t soft_len(int sps, size_t burst_size) { return sps == 4 ? 156 : burst_size; }
size_t soft_cap(int sps, size_t burst_size) { size_t n = soft_len(sps, burst_size); return n < EDGE_BURST_NBITS ? EDGE_BURST_NBITS : n; }

}  // namespace

/* the context for the other host classes of the shim (MultiArfcnRx, BurstGatherer) */
extern "C" trxhip_ctx *trxsigproc_context(void) { return g_ctx; }

/* Further contexts for the multi-device gatherer: the table blob is generated ONCE per process on the host (sigProcLibSetup()'s
 * work, sigProcLib.cpp:2139-2172) and every device gets a plain upload of it -- what SURVEY section 5 allows inside one process
 * where bench.py's one-process-per-GPU form uses the RCCL broadcast. */
extern "C" trxhip_ctx *trxsigproc_create_context(int device)
{
	static std::mutex mu;
	static std::vector<char> blob;
	std::lock_guard<std::mutex> lk(mu);
	if (blob.empty()) {
		blob.resize(trxhip_tables_size());
		if (trxhip_tables_generate_host(blob.data(), blob.size()) != TRXHIP_OK) {
			blob.clear();
			return nullptr;
		}
	}
	trxhip_ctx *ctx = nullptr;
	const int rc = trxhip_create_from_tables(&ctx, device, blob.data(), blob.size());
	if (rc != TRXHIP_OK) {
		fprintf(stderr, "trxsigproc: device %d: %s\n", device, trxhip_strerror(rc));
		return nullptr;
	}
	return ctx;
}
extern "C" void trxsigproc_destroy_context(trxhip_ctx *ctx) { trxhip_destroy(ctx); }

/* Which headers this library was compiled against, and the object layout it therefore assumes: a caller (or
 * tests/test_shim_abi.py) compares these with its own sizeof/offsetof before handing objects across. */
extern "C" const char *trxsigproc_abi(void) { return TRX_SHIM_ABI; }
extern "C" void trxsigproc_abi_layout(size_t out[8])
{
	out[0] = sizeof(signalVector);
	out[1] = sizeof(SoftVector);
	out[2] = sizeof(complex);
	out[3] = sizeof(struct estim_burst_params);
	out[4] = offsetof(struct estim_burst_params, toa);
	out[5] = offsetof(struct estim_burst_params, tsc);
	out[6] = offsetof(struct estim_burst_params, ci);
	out[7] = sizeof(Vector<float>);
}

static void gpu_error(const char *where)
{
	/* the reference would LOG(ERR); the shim has no libosmocore logging context, stderr it is */
	fprintf(stderr, "trxsigproc: %s: GPU error (%s)\n", where, hipGetErrorString(hipGetLastError()));
}

bool sigProcLibSetup()
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_ctx)
		return true;
	const char *dev = getenv("TRXHIP_DEVICE");
	int rc = trxhip_create(&g_ctx, dev ? atoi(dev) : 0);
	if (rc != TRXHIP_OK) {
		fprintf(stderr, "sigProcLibSetup: %s\n", trxhip_strerror(rc));
		g_ctx = nullptr;
		return false;
	}
	return true;
}

void sigProcLibDestroy(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	trxhip_destroy(g_ctx);
	g_ctx = nullptr;
}

void vectorSlicer(float *dest, const float *src, size_t len)
{
	Scratch &t = tls;
	if (!len)
		return;
	if (!g_ctx || !t.ensure(1, 2 * len * sizeof(float), 1)) {
		gpu_error("vectorSlicer");
		return;
	}
	float *d_src = static_cast<float *>(t.d_iq), *d_dst = d_src + len;
	if (!(h2d(d_src, src, len * sizeof(float), t.stream) &&
	      trxhip_vector_slicer(g_ctx, d_dst, d_src, len, t.stream) == TRXHIP_OK &&
	      d2h(dest, d_dst, len * sizeof(float), t.stream) && hipStreamSynchronize(t.stream) == hipSuccess))
		gpu_error("vectorSlicer");
}

float energyDetect(const signalVector &rxBurst, unsigned windowLength)
{
	Scratch &t = tls;
	if (!g_ctx || windowLength == 0 || !t.ensure(1, rxBurst.bytes(), 1))
		return 0.0f;
	float e = 0.0f;
	if (!h2d(t.d_iq, rxBurst.begin(), rxBurst.bytes(), t.stream) ||
	    trxhip_energy_detect_batch_cf32(g_ctx, static_cast<const float *>(t.d_iq), 1, (int)rxBurst.size(), windowLength,
					    t.d_soft, t.stream) != TRXHIP_OK ||
	    !d2h(&e, t.d_soft, sizeof(float), t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)
		return 0.0f;
	return e;
}

int detectAnyBurst(const signalVector &burst, unsigned tsc, float threshold, int sps, CorrType type,
		   unsigned max_toa, struct estim_burst_params *ebp)
{
	Scratch &t = tls;
	t.last_burst = nullptr;
	if ((sps != 1) && (sps != 4))
		return -SIGERR_UNSUPPORTED;                    /* sigProcLib.cpp:1740-1741 */
	if (!g_ctx || !ebp)
		return -SIGERR_INTERNAL;
	const size_t n = burst.size();
	const size_t stride = soft_cap(sps, n);                /* room for 444 8-PSK soft bits */
	if (!t.ensure(1, burst.bytes(), stride))
		return -SIGERR_INTERNAL;

	trxhip_burst_params prm;
	memset(&prm, 0, sizeof(prm));
	prm.type = (uint8_t)type;
	prm.tsc = (uint8_t)(tsc > 255 ? 255 : tsc);
	prm.max_toa = (uint16_t)(max_toa > 65535 ? 65535 : max_toa);
	trxhip_burst_result res;
	t.last_soft.assign(stride, 0.0f);
	if (!h2d(t.d_iq, burst.begin(), burst.bytes(), t.stream) || !h2d(t.d_prm, &prm, sizeof(prm), t.stream))
		return -SIGERR_INTERNAL;
	int rc = trxhip_detect_demod_batch_cf32(g_ctx, static_cast<const float *>(t.d_iq), t.d_prm, t.d_res, t.d_soft, 1,
						(int)n, sps, threshold, 1.0f, (int)stride,
						TRXHIP_FLAG_EXACT_DEMOD /* raw soft bits, reference operand order */ |
						TRXHIP_FLAG_IDLE_DUMMY /* detectAnyBurst(IDLE) = detectDummyBurst, :1945-1947 */, t.stream);
	if (rc != TRXHIP_OK || !d2h(&res, t.d_res, sizeof(res), t.stream) ||
	    !d2h(t.last_soft.data(), t.d_soft, stride * sizeof(float), t.stream) ||
	    hipStreamSynchronize(t.stream) != hipSuccess)
		return -SIGERR_INTERNAL;

	ebp->amp = complex(res.amp_re, res.amp_im);
	ebp->toa = res.toa;
	ebp->ci = res.ci;
	if (type == TSC || type == EDGE || type == IDLE)
		ebp->tsc = (type == IDLE) ? 0 : (uint8_t)tsc;         /* sigProcLib.cpp:1901,1920,1874 */
	if (res.rc > 0 && (type == RACH || type == EXT_RACH))
		ebp->tsc = res.tsc;                                    /* :1797 */
	if (res.rc > 0) {
		t.last_burst = burst.begin();
		t.last_size = n;
		t.last_sps = sps;
		t.last_rc = res.rc;
		t.last_ebp = *ebp;
	}
	return res.rc;
}

signalVector *delayVector(const signalVector *in, signalVector *out, float delay)
{
	Scratch &t = tls;
	if (!g_ctx || !in || !t.ensure(1, 2 * in->bytes() + 16, 1))
		return NULL;
	const size_t n = in->size();
	signalVector *res = new signalVector(n);
	/* scratch layout: [in][out]; the delay rides in d_ebp */
	float *d_in = static_cast<float *>(t.d_iq), *d_out = d_in + 2 * n;
	if (n && (!h2d(d_in, in->begin(), in->bytes(), t.stream) || !h2d(t.d_ebp, &delay, sizeof(float), t.stream) ||
		  trxhip_delay_vector_batch_cf32(g_ctx, d_in, d_out, t.d_ebp, 1, (int)n, t.stream) != TRXHIP_OK ||
		  !d2h(res->begin(), d_out, res->bytes(), t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)) {
		delete res;
		return NULL;
	}
	if (!out)
		return res;
	out->clone(*res);                                          /* :1094 */
	delete res;
	return out;
}

void scaleVector(signalVector &x, complex scale)
{
	Scratch &t = tls;
	if (!x.size())
		return;
	if (!g_ctx || !t.ensure(1, x.bytes(), 1)) {
		gpu_error("scaleVector");
		return;
	}
	const complex *src = x.begin();
	std::vector<complex> re;
	if (x.isReal()) {                                          /* *xP = xP->real() * scale (:1207-1210): imaginary parts ignored */
		re.assign(x.begin(), x.end());
		for (size_t k = 0; k < re.size(); k++)
			re[k] = complex(re[k].real(), 0.0f);
		src = re.data();
	}
	if (!(h2d(t.d_iq, src, x.bytes(), t.stream) &&
	      trxhip_scale_vector_cf32(g_ctx, static_cast<float *>(t.d_iq), x.size(), scale.real(), scale.imag(), t.stream) == TRXHIP_OK &&
	      d2h(x.begin(), t.d_iq, x.bytes(), t.stream) && hipStreamSynchronize(t.stream) == hipSuccess))
		gpu_error("scaleVector");
}

SoftVector *demodAnyBurst_va(const signalVector &burst, CorrType type, int sps, int rach_max_toa, int tsc)
{
	Scratch &t = tls;
	(void)sps;                                                 /* the reference ignores it too: the receiver is 4 SPS */
	if (!g_ctx || tsc < 0 || tsc > 7 || !burst.size() || !t.ensure(1, burst.bytes(), 156))
		return NULL;
	trxhip_burst_params prm;
	memset(&prm, 0, sizeof(prm));
	prm.type = (uint8_t)type;
	prm.tsc = (uint8_t)tsc;
	prm.max_toa = (uint16_t)(rach_max_toa < 0 ? 0 : rach_max_toa > 65535 ? 65535 : rach_max_toa);
	SoftVector *bits = new SoftVector(148 + 8);
	if (!h2d(t.d_iq, burst.begin(), burst.bytes(), t.stream) || !h2d(t.d_prm, &prm, sizeof(prm), t.stream) ||
	    trxhip_demod_va_batch_cf32(g_ctx, static_cast<const float *>(t.d_iq), t.d_prm, NULL, t.d_soft, NULL, 1, (int)burst.size(),
				       1.0f, 156, 0, t.stream) != TRXHIP_OK ||
	    !d2h(bits->begin(), t.d_soft, 156 * sizeof(float), t.stream) || hipStreamSynchronize(t.stream) != hipSuccess) {
		delete bits;
		return NULL;
	}
	return bits;
}

int detectSCHBurst(signalVector &burst, float thresh, int sps, sch_detect_type state, struct estim_burst_params *ebp)
{
	Scratch &t = tls;
	if ((sps != 1) && (sps != 4))
		return -1;                                             /* sigProcLib.cpp:1814-1815 */
	if (!g_ctx || !ebp || !t.ensure(1, burst.bytes(), 1))
		return -1;
	const int st = (state == sch_detect_type::SCH_DETECT_NARROW) ? TRXHIP_SCH_DETECT_NARROW
		     : (state == sch_detect_type::SCH_DETECT_BUFFER) ? TRXHIP_SCH_DETECT_BUFFER : TRXHIP_SCH_DETECT_FULL;
	trxhip_burst_result res;
	if (!h2d(t.d_iq, burst.begin(), burst.bytes(), t.stream) ||
	    trxhip_detect_sch_batch_cf32(g_ctx, static_cast<const float *>(t.d_iq), t.d_res, 1, burst.size(), sps, st, thresh,
					 t.stream) != TRXHIP_OK ||
	    !d2h(&res, t.d_res, sizeof(res), t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)
		return -1;
	/* tsc is left alone, as in the reference (only detectBurst()'s fields are written, :1691-1704, :1846-1858) */
	ebp->amp = complex(res.amp_re, res.amp_im);
	ebp->toa = res.toa;
	if (res.rc > 0)
		ebp->ci = res.ci;
	return res.rc;
}

SoftVector *demodAnyBurst(const signalVector &burst, CorrType type, int sps, struct estim_burst_params *ebp)
{
	Scratch &t = tls;
	if (!g_ctx || !ebp || ((sps != 1) && (sps != 4)))
		return NULL;
	const size_t n = burst.size();
	const size_t ns = (type == EDGE) ? EDGE_BURST_NBITS : soft_len(sps, n);   /* sigProcLib.cpp:1970, :2013 */
	SoftVector *bits = new SoftVector(ns);

	/* the fused kernel already demodulated this burst during detectAnyBurst() with these very parameters */
	if (t.last_burst == burst.begin() && t.last_size == n && t.last_sps == sps && t.last_rc == (int)type &&
	    t.last_ebp.toa == ebp->toa && t.last_ebp.amp == ebp->amp && t.last_soft.size() >= ns) {
		memcpy(bits->begin(), t.last_soft.data(), ns * sizeof(float));
		return bits;                  /* (8-PSK: ebp->ci already holds the EVM estimate of sigProcLib.cpp:2118) */
	}

	const size_t stride = soft_cap(sps, n);
	trxhip_burst_params prm;
	memset(&prm, 0, sizeof(prm));
	prm.type = (uint8_t)type;
	prm.tsc = ebp->tsc;
	const float e[4] = { ebp->toa, ebp->amp.real(), ebp->amp.imag(), 0.0f };
	if (!t.ensure(1, burst.bytes(), stride) || !h2d(t.d_iq, burst.begin(), burst.bytes(), t.stream) ||
	    !h2d(t.d_prm, &prm, sizeof(prm), t.stream) || !h2d(t.d_ebp, e, sizeof(e), t.stream) ||
	    trxhip_demod_batch_cf32(g_ctx, static_cast<const float *>(t.d_iq), t.d_prm, t.d_ebp, t.d_res, t.d_soft, 1, (int)n,
				    sps, (int)stride, TRXHIP_FLAG_EXACT_DEMOD, t.stream) != TRXHIP_OK ||
	    !d2h(bits->begin(), t.d_soft, ns * sizeof(float), t.stream) || !d2h(&t.last_res, t.d_res, sizeof(t.last_res), t.stream) ||
	    hipStreamSynchronize(t.stream) != hipSuccess) {
		delete bits;
		return NULL;
	}
	if (type == EDGE)
		ebp->ci = t.last_res.ci;      /* demodEdgeBurst() replaces C/I by the EVM estimate (sigProcLib.cpp:2118) */
	return bits;
}

void trxsigproc_fill_indication(BurstIndication &bi, const BurstRequest &rq, const trxhip_burst_result &r, const float *soft,
				size_t stride, double rssi_offset)
{
	bi.nbits = 0;
	bi.fn = rq.fn;
	bi.tn = rq.tn;
	bi.rssi = 0.0;
	bi.toa = 0.0;
	bi.idle = r.idle != 0;
	bi.modulation = 0;
	bi.tss = 0;
	bi.tsc = 0;
	bi.ci = 0.0f;
	bi.rc = r.rc;
	bi.energy = r.energy;
	bi.type = (uint8_t)rq.type;
	if (rq.type != OFF)
		bi.rssi = (double)r.rssi + rssi_offset;                /* Transceiver.cpp:751 */
	if (bi.idle)
		return;
	bi.toa = r.toa;
	bi.tsc = r.tsc;
	bi.ci = r.ci;
	bi.nbits = 4u * r.nbits_div4;
	bi.modulation = bi.nbits == EDGE_BURST_NBITS ? 1 : 0;      /* :794-800 */
	if (soft)
		memcpy(bi.rx_burst, soft, (bi.nbits < stride ? bi.nbits : stride) * sizeof(float));
}

int pullRadioVectorBatch(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			 double rssi_offset, BurstIndication *out, bool egprs)
{
	const size_t stride = egprs ? EDGE_BURST_NBITS : NORMAL_BURST_NBITS;
	Scratch &t = tls;
	if (!n)
		return 0;
	if (!g_ctx || !req || !out)
		return -EIO;
	/* pinned staging slots, one stream each: while chunk k's kernels run, chunk k+1 uploads and chunk k-1 downloads.
	 * The pipe is per calling thread and keyed by geometry only (sps, burst length): its slot capacity follows the
	 * largest batch seen (powers of two, 64 .. 2048 bursts: a thread that pulls 8 bursts at a time pins 0.6 MB, not
	 * 26 MB) and its rows the widest ever asked for (444 once an egprs batch came by), both grow-only, so that
	 * alternating batch sizes / egprs settings / full scales never re-allocates; levels are set per call. */
	trxhip_hostpipe_cfg c = t.pipe ? t.pipe_cfg : trxhip_hostpipe_cfg();
	uint32_t cap = 64;
	while (cap < n && cap < 2048)
		cap <<= 1;
	if (!t.pipe || c.sps != sps || c.burst_len != (int32_t)burst_len || c.max_bursts < cap || c.soft_stride < (int32_t)stride) {
		const uint32_t keep_cap = (t.pipe && c.sps == sps && c.burst_len == (int32_t)burst_len) ? c.max_bursts : 0;
		const int32_t keep_stride = t.pipe ? c.soft_stride : 0;
		memset(&c, 0, sizeof(c));
		c.max_bursts = cap > keep_cap ? cap : keep_cap;
		c.depth = 3;
		c.burst_len = (int32_t)burst_len;
		c.sps = sps;
		c.soft_stride = (int32_t)stride > keep_stride ? (int32_t)stride : keep_stride;
		c.flags = TRXHIP_FLAG_SLICE;                           /* vectorSlicer applied; fused demodulator */
		c.threshold = BURST_THRESH;
		c.full_scale = (float)rxFullScale;
		if (t.pipe) trxhip_hostpipe_destroy(t.pipe);
		t.pipe = nullptr;
		if (trxhip_hostpipe_create(g_ctx, &c, &t.pipe) != TRXHIP_OK)
			return -EIO;
		t.pipe_cfg = c;
	}
	if (trxhip_hostpipe_set_levels(t.pipe, BURST_THRESH, (float)rxFullScale, 0.0f) != TRXHIP_OK)
		return -EIO;
	const size_t row = (size_t)c.soft_stride;                  /* the pipe's row width (>= stride) */
	const size_t burst_bytes = burst_len * 2 * sizeof(int16_t);
	const size_t n_chunks = (n + c.max_bursts - 1) / c.max_bursts;
	int err = 0;
	auto collect = [&](size_t k) {
		const int s = (int)(k % c.depth);
		const size_t off = k * c.max_bursts, m = (off + c.max_bursts <= n) ? c.max_bursts : n - off;
		trxhip_hostpipe_slot h;
		if (trxhip_hostpipe_wait(t.pipe, s) != TRXHIP_OK || trxhip_hostpipe_slot_buffers(t.pipe, s, &h) != TRXHIP_OK) {
			err = -EIO;
			return;
		}
		for (size_t i = 0; i < m; i++)
			trxsigproc_fill_indication(out[off + i], req[off + i], h.results[i], h.soft + i * row, stride, rssi_offset);
	};
	for (size_t k = 0; k < n_chunks; k++) {
		if (k >= (size_t)c.depth)
			collect(k - c.depth);
		const int s = (int)(k % c.depth);
		const size_t off = k * c.max_bursts, m = (off + c.max_bursts <= n) ? c.max_bursts : n - off;
		trxhip_hostpipe_slot h;
		if (trxhip_hostpipe_slot_buffers(t.pipe, s, &h) != TRXHIP_OK) { err = -EIO; break; }
		for (size_t i = 0; i < m; i++) {                       /* gather straight into the pinned slot: the only host copy */
			memcpy(h.iq + i * burst_len * 2, req[off + i].iq, burst_bytes);
			memset(&h.params[i], 0, sizeof(h.params[i]));
			h.params[i].type = (uint8_t)req[off + i].type;
			h.params[i].tsc = (uint8_t)req[off + i].tsc;
			h.params[i].max_toa = (uint16_t)req[off + i].max_toa;
		}
		if (trxhip_hostpipe_submit(t.pipe, s, m) != TRXHIP_OK)
			err = -EIO;
	}
	for (size_t k = (n_chunks > (size_t)c.depth ? n_chunks - c.depth : 0); k < n_chunks; k++)
		collect(k);
	for (int s = 0; s < c.depth; s++)
		(void)trxhip_hostpipe_wait(t.pipe, s);
	return err;
}

int pullRadioVectorBatchVA(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			   double rssi_offset, BurstIndication *out)
{
	const size_t stride = NORMAL_BURST_NBITS;
	Scratch &t = tls;
	if (!n)
		return 0;
	if (!g_ctx || !req || !out || burst_len <= 40)
		return -EIO;
	const size_t burst_bytes = burst_len * 2 * sizeof(int16_t);
	/* scratch: [int16 bursts][int16 shifted copy][complex64 bursts]; the energies ride in d_ebp */
	if (!t.ensure(n, 2 * n * burst_bytes + n * burst_len * 2 * sizeof(float), stride))
		return -EIO;
	int16_t *d_iq = static_cast<int16_t *>(t.d_iq), *d_shift = d_iq + n * burst_len * 2;
	float *d_cf = reinterpret_cast<float *>(d_shift + n * burst_len * 2);

	std::vector<int16_t> iq(n * burst_len * 2);
	std::vector<trxhip_burst_params> prm(n);
	for (size_t i = 0; i < n; i++) {
		memcpy(&iq[i * burst_len * 2], req[i].iq, burst_bytes);
		memset(&prm[i], 0, sizeof(prm[i]));
		prm[i].type = (uint8_t)req[i].type;
		prm[i].tsc = (uint8_t)req[i].tsc;
		prm[i].max_toa = (uint16_t)req[i].max_toa;
	}
	std::vector<trxhip_burst_result> res(n);
	std::vector<float> soft(n * stride), energy(n);
	const size_t shift_bytes = (burst_len - 40) * 2 * sizeof(int16_t);
	if (!h2d(d_iq, iq.data(), n * burst_bytes, t.stream) || !h2d(t.d_prm, prm.data(), n * sizeof(prm[0]), t.stream) ||
	    /* shift_vec: samples 20 .. len-20 in front, zeros behind (Transceiver.cpp:679, :762) */
	    hipMemsetAsync(d_shift, 0, n * burst_bytes, t.stream) != hipSuccess ||
	    hipMemcpy2DAsync(d_shift, burst_bytes, d_iq + 40, burst_bytes, shift_bytes, n, hipMemcpyDeviceToDevice, t.stream) != hipSuccess)
		return -EIO;
	int rc = trxhip_detect_demod_batch(g_ctx, d_shift, t.d_prm, t.d_res, NULL, n, (int)burst_len, sps, BURST_THRESH,
					   (float)rxFullScale, (int)stride, 0, t.stream);
	if (rc == TRXHIP_OK)
		rc = trxhip_convert_short_float(g_ctx, d_cf, d_iq, n * burst_len * 2, t.stream);
	if (rc == TRXHIP_OK)                                       /* power of the unshifted burst (Transceiver.cpp:724-746) */
		rc = trxhip_energy_detect_batch_cf32(g_ctx, d_cf, n, (int)burst_len, 20 * sps, t.d_ebp, t.stream);
	if (rc == TRXHIP_OK)
		rc = trxhip_demod_va_batch_cf32(g_ctx, d_cf, t.d_prm, t.d_res, t.d_soft, NULL, n, (int)burst_len,
						(float)(1. / (float)((1 << 14) - 1)), (int)stride, TRXHIP_FLAG_SLICE, t.stream);
	if (rc != TRXHIP_OK || !d2h(res.data(), t.d_res, n * sizeof(res[0]), t.stream) ||
	    !d2h(soft.data(), t.d_soft, soft.size() * sizeof(float), t.stream) ||
	    !d2h(energy.data(), t.d_ebp, n * sizeof(float), t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)
		return -EIO;

	for (size_t i = 0; i < n; i++) {
		BurstIndication &bi = out[i];
		memset(&bi, 0, sizeof(bi));
		bi.fn = req[i].fn;
		bi.tn = req[i].tn;
		bi.rc = res[i].rc;
		bi.idle = res[i].idle != 0;
		bi.energy = energy[i];
		if (req[i].type == OFF)
			bi.energy = 0.0f;                              /* no power level for a slot that is off (Transceiver.cpp:704-707) */
		else                                               /* avg = sqrt(pow / chans); 20 log10(fullscale / avg) (:741, :751) */
			bi.rssi = 20.0 * log10(rxFullScale / (double)sqrtf(energy[i])) + rssi_offset;
		if (!bi.idle) {
			bi.toa = res[i].toa;
			bi.tsc = res[i].tsc;
			bi.ci = res[i].ci;
			bi.nbits = NORMAL_BURST_NBITS;                 /* rxBurst->size() = 156, not 444 (Transceiver.cpp:794-800) */
			memcpy(bi.rx_burst, &soft[i * stride], NORMAL_BURST_NBITS * sizeof(float));
		}
	}
	return 0;
}


TRX_SHIM_NS_END

#ifdef TRX_SHIM_REFERENCE_ABI
/* The same functions under names of their own, for binaries that keep the reference's sigProcLib.o and interpose the
 * receive side at link time (trxWrap.h, trxwrap.cpp).  The library is linked -Bsymbolic-functions: the
 * calls below bind to this library's definitions even when the executable defines functions of the same name. */
#include "trxWrap.h"
namespace trxgpu {
bool sigProcLibSetup() { return ::sigProcLibSetup(); }
void sigProcLibDestroy() { ::sigProcLibDestroy(); }
int detectAnyBurst(const signalVector &burst, unsigned tsc, float threshold, int sps, CorrType type, unsigned max_toa,
		   struct estim_burst_params *ebp)
{
	return ::detectAnyBurst(burst, tsc, threshold, sps, type, max_toa, ebp);
}
SoftVector *demodAnyBurst(const signalVector &burst, CorrType type, int sps, struct estim_burst_params *ebp)
{
	return ::demodAnyBurst(burst, type, sps, ebp);
}
float energyDetect(const signalVector &rxBurst, unsigned windowLength) { return ::energyDetect(rxBurst, windowLength); }
void vectorSlicer(float *dest, const float *src, size_t len) { ::vectorSlicer(dest, src, len); }
signalVector *delayVector(const signalVector *in, signalVector *out, float delay) { return ::delayVector(in, out, delay); }
void scaleVector(signalVector &x, complex scale) { ::scaleVector(x, scale); }
int detectSCHBurst(signalVector &burst, float thresh, int sps, sch_detect_type state, struct estim_burst_params *ebp)
{
	return ::detectSCHBurst(burst, thresh, sps, state, ebp);
}
}
#endif
