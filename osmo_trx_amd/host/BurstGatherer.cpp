// BurstGatherer.cpp -- the gather stage between the reference's per-channel burst FIFOs and the batched GPU path,
// plus the host-side TRXD packer.  See trxBatch.h.
//
// Reference semantics kept (radioInterface.cpp:272-291, Transceiver.cpp:665-815, :1229-1253):
//   * producer never blocks on a slow consumer: a channel with `fifo_depth` (32) bursts pushed and not yet pulled
//     drops the new burst (radioInterface.cpp:277-280);
//   * consumer blocks until "its" next burst is there and gets exactly one indication per pushed burst, in push order,
//     with pullRadioVector()'s return codes (0 / -ENOENT for an OFF slot / -EIO);
//   * nothing is reordered across the channel boundary that the reference would keep ordered: channels are independent.
// New: bursts of all channels are gathered into one staging batch (written by the producers themselves, straight
// into pinned memory -- no second copy) that is launched when it holds max_batch bursts or its first burst has
// waited timeout_us; `depth` batches are in flight on their own streams (trxhip_hostpipe_*), so upload, kernels and
// download of consecutive batches overlap.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include <cmath>

#include "shim_internal.h"

TRX_SHIM_NS_BEGIN

/* proto_trxd.c:28-117, restated for one indication on the host (the batched form is pack_trxd_wire_kernel) */
int trxdPackBurstInd(uint8_t *buf, const BurstIndication *bi, unsigned version)
{
	if (version > 1)
		return -1;
	if (version == 0 && bi->idle)
		return 0;                                              /* v0 doesn't support idle frames (:71-73) */
	const unsigned nbits = bi->idle ? 0 : bi->nbits;
	buf[0] = (uint8_t)(((version & 0xf) << 4) | (bi->tn & 7));         /* trxd_fill_common() */
	buf[1] = (uint8_t)(bi->fn >> 24); buf[2] = (uint8_t)(bi->fn >> 16); buf[3] = (uint8_t)(bi->fn >> 8); buf[4] = (uint8_t)bi->fn;
	/* v0->rssi = bi->rssi: double -> uint8_t, defined for 0..255 only; saturate outside (NaN -> 0) */
	buf[5] = bi->rssi >= 255.0 ? 255 : (bi->rssi > 0.0 ? (uint8_t)bi->rssi : 0);
	const int toa_int = (int)(bi->toa * 256.0 + 0.5);                  /* trxd_fill_v0_specific() */
	buf[6] = (uint8_t)((unsigned)toa_int >> 8); buf[7] = (uint8_t)toa_int;
	size_t pos = TRXD_V0_HDR_LEN;
	if (version == 1) {                                                /* trxd_fill_v1_specific() */
		const int16_t ci_cb = (int16_t)((bi->ci * 10) + 0.5);
		const unsigned mod = bi->modulation == 0 ? (0u | (bi->tss & 3u)) : (4u | (bi->tss & 1u));
		buf[8] = (uint8_t)(((bi->idle ? 1u : 0u) << 7) | (mod << 3) | (bi->tsc & 7u));
		buf[9] = (uint8_t)((uint16_t)ci_cb >> 8); buf[10] = (uint8_t)ci_cb;
		pos = TRXD_V1_HDR_LEN;
	}
	for (unsigned i = 0; i < nbits; i++)                               /* trxd_fill_burst_normalized255() */
		buf[pos + i] = (uint8_t)round(bi->rx_burst[i] * 255.0);
	pos += nbits;
	if (version == 0) {                                                /* two historical trailing bytes (:76, :83-87) */
		buf[pos++] = 0;
		buf[pos++] = 0;
	}
	return (int)pos;
}

namespace {
typedef std::chrono::steady_clock Clock;

struct Route {
	uint16_t chan;
	uint8_t type;
	uint8_t tn;
	uint32_t fn;
};
struct Done {
	BurstIndication bi;
	int code;
	uint16_t pkt_len;
	uint8_t pkt[TRXD_MAX_PKT_LEN + 1];
};
struct Chan {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<Done> ring;
	size_t head = 0, count = 0;
	std::atomic<size_t> outstanding{0};
};
struct Batch {
	trxhip_hostpipe_slot h;
	uint32_t reserved = 0;                  /* indices handed to producers (under Impl::mu) */
	std::atomic<uint32_t> written{0};       /* producers done copying */
	uint32_t count = 0;                     /* final size once closed */
	Clock::time_point first;
	std::vector<Route> route;
};
}  // namespace

struct BurstGatherer::Impl {
	BurstGathererConfig cfg;
	trxhip_hostpipe *pipe = nullptr;
	size_t stride = 0;
	std::vector<Batch> batch;
	std::vector<Chan> chan;
	std::mutex mu;
	std::condition_variable cv_work, cv_space, cv_done;
	int filling = -1;
	std::deque<int> free_q, closed_q, flight_q;
	std::atomic<bool> stopping{false}, running{false}, failed{false};
	std::thread submitter, completer;
	std::atomic<uint64_t> n_batches{0}, n_dropped{0};

	void close_filling_locked()
	{
		Batch &b = batch[filling];
		b.count = b.reserved;
		closed_q.push_back(filling);
		filling = -1;
		if (!free_q.empty()) {
			filling = free_q.front();
			free_q.pop_front();
		}
		cv_work.notify_all();
	}

	void submit_loop()
	{
		std::unique_lock<std::mutex> lk(mu);
		while (!stopping) {
			if (!closed_q.empty()) {
				const int s = closed_q.front();
				closed_q.pop_front();
				Batch &b = batch[s];
				lk.unlock();
				while (b.written.load(std::memory_order_acquire) < b.count)   /* a producer is still inside its memcpy */
					std::this_thread::yield();
				const int rc = trxhip_hostpipe_submit(pipe, s, b.count);
				lk.lock();
				if (rc != TRXHIP_OK)
					failed = true;
				n_batches++;
				flight_q.push_back(s);
				cv_done.notify_all();
				continue;
			}
			if (filling >= 0 && batch[filling].reserved > 0) {
				const Clock::time_point deadline = batch[filling].first + std::chrono::microseconds(cfg.timeout_us);
				if (Clock::now() >= deadline) {
					close_filling_locked();
					continue;
				}
				cv_work.wait_until(lk, deadline);
			} else {
				cv_work.wait(lk);
			}
		}
	}

	void deliver(size_t c, const Done &d)
	{
		Chan &ch = chan[c];
		std::lock_guard<std::mutex> g(ch.mu);
		/* cannot overflow: outstanding <= fifo_depth = ring size */
		ch.ring[(ch.head + ch.count) % ch.ring.size()] = d;
		ch.count++;
		ch.cv.notify_one();
	}

	void complete_loop()
	{
		std::unique_lock<std::mutex> lk(mu);
		Done d;
		for (;;) {
			cv_done.wait(lk, [&] { return stopping || !flight_q.empty(); });
			if (flight_q.empty())
				return;                                        /* stopping and drained */
			const int s = flight_q.front();
			flight_q.pop_front();
			Batch &b = batch[s];
			lk.unlock();
			const bool ok = trxhip_hostpipe_wait(pipe, s) == TRXHIP_OK;
			for (uint32_t i = 0; i < b.count; i++) {
				const Route &r = b.route[i];
				BurstRequest rq;
				memset(&rq, 0, sizeof(rq));
				rq.type = (CorrType)r.type;
				rq.fn = r.fn;
				rq.tn = r.tn;
				d.code = !ok ? -EIO : (r.type == OFF ? -ENOENT : 0);
				d.pkt_len = 0;
				if (ok) {
					trxsigproc_fill_indication(d.bi, rq, b.h.results[i], b.h.soft ? b.h.soft + i * stride : NULL, stride,
								   cfg.rssi_offset);
					if (b.h.pkt) {
						d.pkt_len = b.h.pkt_len[i];
						memcpy(d.pkt, b.h.pkt + (size_t)i * stride, d.pkt_len);
					}
				} else {
					memset(&d.bi, 0, sizeof(d.bi));
					d.bi.fn = r.fn;
					d.bi.tn = r.tn;
				}
				deliver(r.chan, d);
			}
			lk.lock();
			b.reserved = 0;
			b.written.store(0, std::memory_order_relaxed);
			b.count = 0;
			if (filling < 0)
				filling = s;
			else
				free_q.push_back(s);
			cv_space.notify_all();
		}
	}
};

BurstGatherer::BurstGatherer(const BurstGathererConfig &cfg) : impl_(new Impl())
{
	impl_->cfg = cfg;
	if (impl_->cfg.fifo_depth == 0) impl_->cfg.fifo_depth = 32;
	if (impl_->cfg.depth < 2) impl_->cfg.depth = 2;
	if (impl_->cfg.depth > 16) impl_->cfg.depth = 16;
	if (impl_->cfg.max_batch == 0) impl_->cfg.max_batch = 128;
}

BurstGatherer::~BurstGatherer()
{
	stop();
	if (impl_->pipe)
		trxhip_hostpipe_destroy(impl_->pipe);
	delete impl_;
}

bool BurstGatherer::start()
{
	Impl &m = *impl_;
	if (m.running || !trxsigproc_context() || m.cfg.chans == 0 || m.cfg.chans > 65535)
		return false;
	trxhip_hostpipe_cfg c;
	memset(&c, 0, sizeof(c));
	c.max_bursts = (uint32_t)m.cfg.max_batch;
	c.depth = m.cfg.depth;
	c.burst_len = (int32_t)m.cfg.burst_len;
	c.sps = m.cfg.sps;
	if (m.cfg.trxd_version < 0) {
		c.soft_stride = m.cfg.egprs ? EDGE_BURST_NBITS : NORMAL_BURST_NBITS;
		m.stride = c.soft_stride;
	} else {
		if (m.cfg.trxd_version > 1)
			return false;
		c.pkt_stride = m.cfg.egprs ? 456 : 160;
		m.stride = c.pkt_stride;
	}
	c.flags = TRXHIP_FLAG_SLICE;
	c.threshold = BURST_THRESH;
	c.full_scale = (float)m.cfg.rxFullScale;
	c.rssi_offset = (float)m.cfg.rssi_offset;
	if (trxhip_hostpipe_create(trxsigproc_context(), &c, &m.pipe) != TRXHIP_OK)
		return false;
	m.batch = std::vector<Batch>(m.cfg.depth);
	for (int s = 0; s < m.cfg.depth; s++) {
		trxhip_hostpipe_slot_buffers(m.pipe, s, &m.batch[s].h);
		m.batch[s].route.resize(m.cfg.max_batch);
		if (s)
			m.free_q.push_back(s);
	}
	m.filling = 0;
	m.chan = std::vector<Chan>(m.cfg.chans);
	for (size_t c2 = 0; c2 < m.cfg.chans; c2++)
		m.chan[c2].ring.resize(m.cfg.fifo_depth);
	m.stopping = false;
	m.running = true;
	m.submitter = std::thread([&m] { m.submit_loop(); });
	m.completer = std::thread([&m] { m.complete_loop(); });
	return true;
}

void BurstGatherer::stop()
{
	Impl &m = *impl_;
	if (!m.running)
		return;
	{
		std::lock_guard<std::mutex> g(m.mu);
		m.stopping = true;
		m.cv_work.notify_all();
		m.cv_done.notify_all();
		m.cv_space.notify_all();
	}
	m.submitter.join();
	m.completer.join();
	for (size_t c = 0; c < m.chan.size(); c++) {
		std::lock_guard<std::mutex> g(m.chan[c].mu);
		m.chan[c].cv.notify_all();
	}
	m.running = false;
}

bool BurstGatherer::push(size_t c, const BurstRequest &rq)
{
	Impl &m = *impl_;
	if (c >= m.chan.size() || !rq.iq)
		return false;
	Batch *b;
	uint32_t idx;
	{
		std::unique_lock<std::mutex> lk(m.mu);
		if (m.stopping || !m.running)
			return false;
		if (m.chan[c].outstanding.load(std::memory_order_relaxed) >= m.cfg.fifo_depth) {
			m.n_dropped++;                                         /* radioInterface.cpp:277-280: FIFO full, burst deleted */
			return false;
		}
		/* all staging batches closed or in flight: only possible when depth * max_batch < chans * fifo_depth */
		m.cv_space.wait(lk, [&] { return m.stopping || m.filling >= 0; });
		if (m.stopping)
			return false;
		b = &m.batch[m.filling];
		idx = b->reserved++;
		if (idx == 0) {
			b->first = Clock::now();
			m.cv_work.notify_all();                                /* arm the timeout */
		}
		Route &r = b->route[idx];
		r.chan = (uint16_t)c;
		r.type = (uint8_t)rq.type;
		r.tn = rq.tn;
		r.fn = rq.fn;
		m.chan[c].outstanding.fetch_add(1, std::memory_order_relaxed);
		if (b->reserved == m.cfg.max_batch)
			m.close_filling_locked();
	}
	/* outside the lock: the burst goes straight into the pinned slot the DMA engine reads from */
	memcpy(b->h.iq + (size_t)idx * m.cfg.burst_len * 2, rq.iq, m.cfg.burst_len * 2 * sizeof(int16_t));
	trxhip_burst_params &p = b->h.params[idx];
	p.type = (uint8_t)rq.type;
	p.tsc = (uint8_t)rq.tsc;
	p.max_toa = (uint16_t)rq.max_toa;
	p.reserved = 0;
	if (b->h.meta) {
		trxhip_trxd_meta &mt = b->h.meta[idx];
		mt.fn = rq.fn;
		mt.tn = rq.tn;
		mt.version = (uint8_t)m.cfg.trxd_version;
		mt.tss = 0;
		mt.reserved = 0;
	}
	b->written.fetch_add(1, std::memory_order_release);
	return true;
}

int BurstGatherer::pull(size_t c, BurstIndication *bi, uint8_t *pkt, size_t *pkt_len)
{
	Impl &m = *impl_;
	if (c >= m.chan.size() || !bi)
		return -EIO;
	Chan &ch = m.chan[c];
	std::unique_lock<std::mutex> lk(ch.mu);
	ch.cv.wait(lk, [&] { return ch.count > 0 || m.stopping || !m.running; });
	if (ch.count == 0)
		return -EIO;
	const Done &d = ch.ring[ch.head];
	if (m.cfg.trxd_version < 0 || !pkt) {
		*bi = d.bi;
	} else {                                                       /* header fields only: the soft bits are in the datagram */
		memcpy(reinterpret_cast<char *>(bi) + sizeof(bi->rx_burst), reinterpret_cast<const char *>(&d.bi) + sizeof(bi->rx_burst),
		       sizeof(*bi) - sizeof(bi->rx_burst));
		memcpy(pkt, d.pkt, d.pkt_len);
	}
	if (pkt_len)
		*pkt_len = d.pkt_len;
	const int code = d.code;
	ch.head = (ch.head + 1) % ch.ring.size();
	ch.count--;
	ch.outstanding.fetch_sub(1, std::memory_order_relaxed);
	return code;
}

uint64_t BurstGatherer::batches() const { return impl_->n_batches.load(); }
uint64_t BurstGatherer::dropped() const { return impl_->n_dropped.load(); }

TRX_SHIM_NS_END
