// hostpipe_stub.cpp -- CPU stand-in for the trxhip_hostpipe_* C ABI (include/trxhip.h) and the two shim symbols
// BurstGatherer.cpp needs, so that the gather stage can run under ThreadSanitizer / AddressSanitizer without a GPU.
// TEST INFRASTRUCTURE ONLY: never linked into the product.  The "GPU" echoes each burst's routing stamp back through
// the result record after a random delay, which is all the ordering / exactly-once checks need:
//   iq[0] = fn & 0x7fff, iq[1] = fn >> 15, iq[2] = channel   ->   result.toa = fn (exact below 2^24), result.amp_re = channel
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "shim_internal.h"

struct trxhip_hostpipe {
	trxhip_ctx *ctx;
	trxhip_hostpipe_cfg cfg;
	std::vector<trxhip_hostpipe_slot> slot;
	std::vector<std::atomic<int64_t>> done_ns;
	std::vector<void *> allocs;
	std::vector<const int16_t **> src;                       /* by reference: one pointer array per slot */
	std::vector<std::pair<const char *, size_t>> ranges;
};

static std::atomic<int> g_live_pipes{0};
extern "C" int stub_live_pipes(void) { return g_live_pipes.load(); }

static int64_t now_ns()
{
	return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" trxhip_ctx *trxsigproc_context(void) { return reinterpret_cast<trxhip_ctx *>(0x1000); }
/* fake devices of the multi-device gatherer: a context is an address that encodes its device; the stub counts the live ones
 * and the submits each device saw */
static std::atomic<int> g_live_ctx{0};
static std::atomic<long> g_submits_on[64];
extern "C" int stub_live_contexts(void) { return g_live_ctx.load(); }
extern "C" long stub_submits_on(int device) { return g_submits_on[device & 63].load(); }
extern "C" trxhip_ctx *trxsigproc_create_context(int device)
{
	if (device < 0 || device >= 64)
		return nullptr;
	g_live_ctx++;
	return reinterpret_cast<trxhip_ctx *>((uintptr_t)0x100000 + (uintptr_t)device * 16);
}
extern "C" void trxsigproc_destroy_context(trxhip_ctx *ctx) { if (ctx) g_live_ctx--; }
static int device_of(const trxhip_ctx *ctx)
{
	const uintptr_t a = reinterpret_cast<uintptr_t>(ctx);
	return a >= 0x100000 ? (int)((a - 0x100000) / 16) : 63;        /* 63: the sigProcLibSetup() context */
}

extern "C" int trxhip_hostpipe_create(trxhip_ctx *ctx, const trxhip_hostpipe_cfg *cfg, trxhip_hostpipe **out)
{
	trxhip_hostpipe *p = new trxhip_hostpipe();
	p->ctx = ctx;
	p->cfg = *cfg;
	p->slot.resize(cfg->depth);
	p->done_ns = std::vector<std::atomic<int64_t>>(cfg->depth);
	auto grab = [&](size_t bytes) { void *m = calloc(1, bytes ? bytes : 1); p->allocs.push_back(m); return m; };
	for (int s = 0; s < cfg->depth; s++) {
		trxhip_hostpipe_slot &h = p->slot[s];
		h.iq = (int16_t *)grab((size_t)cfg->max_bursts * cfg->burst_len * 2 * sizeof(int16_t));
		h.params = (trxhip_burst_params *)grab(cfg->max_bursts * sizeof(trxhip_burst_params));
		h.results = (trxhip_burst_result *)grab(cfg->max_bursts * sizeof(trxhip_burst_result));
		h.meta = cfg->pkt_stride ? (trxhip_trxd_meta *)grab(cfg->max_bursts * sizeof(trxhip_trxd_meta)) : nullptr;
		/* exactly the row widths the caller configured: ASan sees any copy beyond them */
		h.soft = cfg->soft_stride ? (float *)grab((size_t)cfg->max_bursts * cfg->soft_stride * sizeof(float)) : nullptr;
		h.pkt = cfg->pkt_stride ? (uint8_t *)grab((size_t)cfg->max_bursts * cfg->pkt_stride) : nullptr;
		h.pkt_len = (uint16_t *)grab(cfg->max_bursts * sizeof(uint16_t));
		p->done_ns[s].store(0);
		p->src.push_back((const int16_t **)grab(cfg->max_bursts * sizeof(void *)));
	}
	g_live_pipes++;
	*out = p;
	return TRXHIP_OK;
}

extern "C" void trxhip_hostpipe_destroy(trxhip_hostpipe *p)
{
	if (!p) return;
	for (void *m : p->allocs) free(m);
	g_live_pipes--;
	delete p;
}

extern "C" int trxhip_hostpipe_slot_buffers(trxhip_hostpipe *p, int slot, trxhip_hostpipe_slot *out)
{
	if (!p || slot < 0 || slot >= p->cfg.depth) return TRXHIP_EINVAL;
	*out = p->slot[slot];
	return TRXHIP_OK;
}

extern "C" int trxhip_hostpipe_submit(trxhip_hostpipe *p, int slot, size_t n)
{
	static thread_local std::mt19937 rng(12345);
	trxhip_hostpipe_slot &h = p->slot[slot];
	g_submits_on[device_of(p->ctx) & 63]++;
	for (size_t i = 0; i < n; i++) {
		const int16_t *iq = h.iq + i * p->cfg.burst_len * 2;
		trxhip_burst_result &r = h.results[i];
		memset(&r, 0, sizeof(r));
		const uint32_t fn = (uint32_t)(uint16_t)iq[0] | ((uint32_t)(uint16_t)iq[1] << 15);
		const bool edge = h.params[i].type == TRXHIP_EDGE;
		r.rc = h.params[i].type;
		r.toa = (float)fn;
		r.amp_re = (float)iq[2];
		r.idle = (h.params[i].type == TRXHIP_OFF || h.params[i].type == TRXHIP_IDLE);
		r.nbits_div4 = r.idle ? 0 : (edge ? 111 : 37);             /* the kernels report 444 bits whatever the row width */
		r.tsc = h.params[i].tsc;
		if (h.soft)
			for (int k = 0; k < p->cfg.soft_stride; k++)
				h.soft[i * p->cfg.soft_stride + k] = (float)((fn + k) & 1);
		if (h.pkt) {
			uint8_t *pk = h.pkt + i * p->cfg.pkt_stride;
			memset(pk, 0, p->cfg.pkt_stride);
			pk[0] = (uint8_t)((h.meta[i].version << 4) | (h.meta[i].tn & 7));
			pk[1] = (uint8_t)(h.meta[i].fn >> 24); pk[2] = (uint8_t)(h.meta[i].fn >> 16);
			pk[3] = (uint8_t)(h.meta[i].fn >> 8);  pk[4] = (uint8_t)h.meta[i].fn;
			h.pkt_len[i] = (uint16_t)(h.meta[i].version ? 11 + 148 : 8 + 148 + 2);
		}
	}
	p->done_ns[slot].store(now_ns() + 20000 + (int64_t)(rng() % 280000), std::memory_order_release);
	return TRXHIP_OK;
}

/* bursts by reference: the "device" copies each burst from its registered range into the slot, then the plain submit */
extern "C" int trxhip_hostpipe_register_host(trxhip_hostpipe *p, const void *base, size_t bytes)
{
	if (!p || !base || !bytes || p->ranges.size() >= 8) return TRXHIP_EINVAL;
	p->ranges.emplace_back((const char *)base, bytes);
	return TRXHIP_OK;
}
extern "C" int trxhip_hostpipe_unregister_host(trxhip_hostpipe *p, const void *base)
{
	for (size_t k = 0; p && k < p->ranges.size(); k++)
		if (p->ranges[k].first == base) { p->ranges.erase(p->ranges.begin() + k); return TRXHIP_OK; }
	return TRXHIP_EINVAL;
}
extern "C" int trxhip_hostpipe_slot_sources(trxhip_hostpipe *p, int slot, const int16_t ***out)
{
	if (!p || !out || slot < 0 || slot >= p->cfg.depth) return TRXHIP_EINVAL;
	*out = p->src[slot];
	return TRXHIP_OK;
}
extern "C" int trxhip_hostpipe_submit_by_ref(trxhip_hostpipe *p, int slot, size_t n)
{
	const size_t bytes = (size_t)p->cfg.burst_len * 4;
	for (size_t i = 0; i < n; i++) {
		const char *q = (const char *)p->src[slot][i];
		bool in = false;
		for (auto &r : p->ranges)
			in = in || (q >= r.first && q + bytes <= r.first + r.second);
		if (!in) return TRXHIP_EINVAL;
	}
	for (size_t i = 0; i < n; i++)
		memcpy(p->slot[slot].iq + i * p->cfg.burst_len * 2, p->src[slot][i], bytes);
	return trxhip_hostpipe_submit(p, slot, n);
}

extern "C" int trxhip_hostpipe_wait(trxhip_hostpipe *p, int slot)
{
	const int64_t t = p->done_ns[slot].load(std::memory_order_acquire);
	while (now_ns() < t)
		std::this_thread::sleep_for(std::chrono::microseconds(20));
	return TRXHIP_OK;
}

TRX_SHIM_NS_BEGIN
void trxsigproc_fill_indication(BurstIndication &bi, const BurstRequest &rq, const trxhip_burst_result &r, const float *soft,
				size_t stride, double rssi_offset)
{
	bi.fn = rq.fn; bi.tn = rq.tn; bi.rc = r.rc; bi.idle = r.idle != 0; bi.toa = r.toa; bi.tsc = r.tsc; bi.ci = r.ci;
	bi.energy = r.amp_re;                                          /* the stub's channel stamp */
	bi.type = (uint8_t)rq.type;
	bi.rssi = r.rssi + rssi_offset;
	bi.nbits = bi.idle ? 0 : 4u * r.nbits_div4;
	bi.modulation = bi.nbits == EDGE_BURST_NBITS ? 1 : 0;
	if (soft && !bi.idle)
		memcpy(bi.rx_burst, soft, (bi.nbits < stride ? bi.nbits : stride) * sizeof(float));
}
TRX_SHIM_NS_END

/* the stall hook of BurstGatherer::pushSlot() (between reading `filling` and the reservation): about one push in 3000
 * sleeps for 2 ms, i.e. several batch round trips of this stub -- the window of the stale-reservation race */
extern "C" void gatherer_test_stall(void)
{
	static thread_local std::mt19937 rng((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
	if (rng() % 3000 == 0)
		std::this_thread::sleep_for(std::chrono::milliseconds(2));
}
