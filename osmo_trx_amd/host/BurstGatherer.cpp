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
//
// Reservation protocol (the only shared write of a push is one fetch_add on the filling batch's `reserved` word):
//   * exactly one batch at a time is OPEN (reserved < max_batch): the one `filling` names.  Every other batch -- closed and
//     waiting for the submitter, in flight, or parked in free_q -- holds reserved >= CLOSED, so a producer that read
//     `filling`, was descheduled for a whole batch round trip and only then executes its fetch_add lands on a closed word
//     and retries with the current `filling`.  A batch is re-opened (epoch bumped, first_ns cleared, reserved = 0) only at
//     the moment it is published as `filling`, under `mu`.  Hence a reservation always lands in the open batch, batches are
//     closed, submitted and completed in the order they were opened, and a channel's bursts come back in push order.
//   * stop() joins the submitter, lets the completion thread drain what was submitted, and discards bursts that were
//     gathered but not submitted (their pull() returns -EIO, as after any stop); start() after stop() begins from empty
//     FIFOs.  stop() may race with push()/pull() and with another stop() (life_mu); so may start(): every push / pull counts
//     itself in `users` on entry, start() raises `reconfig`, waits for the count to reach zero and only then frees the
//     previous run's pinned slots and rings (entrants that see `reconfig` leave at once: push -> refused, pull -> -EIO).
//     The destructor may not race with anything.
//
// Multi-device dispatch (BurstGathererConfig::devices[], or TRXHIP_DEVICES=...): staging batch b belongs to device entry
// b % n and is that entry's host-pipe slot b / n; batches are opened in the rotating order 0, 1, 2, ... so consecutive
// batches go to consecutive GPUs, and the completion thread waits for them in submission order -- which GPU ran a batch
// never shows in the order a channel sees.  Every entry has its own trxhip_ctx (same tables), pinned slots and streams.
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
inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now().time_since_epoch()).count(); }
/* condition_variable::wait_for() on the steady clock is pthread_cond_clockwait(), which gcc 11's ThreadSanitizer does not
 * intercept (it then believes the mutex stays held across the wait and reports double locks and races between holders of
 * the same mutex).  Under TSan only, wait on the system clock: pthread_cond_timedwait(), which it does model. */
template <typename Rep, typename Period>
inline void timed_wait(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, std::chrono::duration<Rep, Period> d)
{
#ifdef __SANITIZE_THREAD__
	cv.wait_until(lk, std::chrono::system_clock::now() + d);
#else
	cv.wait_for(lk, d);
#endif
}

struct Route {
	uint16_t chan;
	uint8_t type;
	uint8_t tn;
	uint32_t fn;
};
struct Slot {                                   /* routing of one gathered burst + its "copied in" flag */
	Route r;
	std::atomic<uint32_t> ready;                /* == the batch's epoch once the producer has finished writing the slot */
};
/* one delivered burst in a channel's ring: the C ABI's result record, routing fields and the payload the consumer
 * asked for (soft floats or the TRXD datagram) -- a few hundred bytes, not a whole BurstIndication */
struct Entry {
	trxhip_burst_result res;
	Route route;
	int32_t code;
	uint32_t pkt_len;
};
/* One channel's delivery ring: single producer (the completion thread), single consumer (that channel's RxUpper thread,
 * Transceiver.cpp:1229-1253).  The consumer takes entries without a lock -- `tail` is published with release semantics
 * after the entries are written -- and only goes to the mutex / condition variable when the ring is empty. */
struct Chan {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<uint8_t> ring;                  /* fifo_depth x (sizeof(Entry) + payload) */
	alignas(64) std::atomic<size_t> tail{0};    /* entries delivered so far (the completion thread whose turn it is) */
	std::atomic<uint64_t> deliver_seq{0};       /* sequence number of the batch that may write this ring next (several completers) */
	std::atomic<int> waiting{0};                /* the consumer sleeps on cv (checked by the completion thread after publishing) */
	alignas(64) size_t head = 0;                /* entries pulled so far (consumer's own) */
	alignas(64) std::atomic<size_t> outstanding{0};     /* pushed and not yet pulled: the reference's FIFO occupancy */
	std::atomic<int> trxd_version{0};           /* TRXD header version of this channel (mVersionTRXD[chan], Transceiver.cpp:1238) */
};
const uint64_t CLOSED = 1ull << 62;             /* far above any sum of stale fetch_adds */
struct Batch {
	/* the one word every push touches sits alone in its cache line */
	alignas(64) std::atomic<uint64_t> reserved{CLOSED}; /* slots handed out; >= CLOSED unless this is the open (filling) batch */
	alignas(64) std::atomic<int64_t> first_ns{0};       /* arrival of the first burst (0 = none yet) */
	trxhip_hostpipe_slot h;
	uint32_t count = 0;                         /* final size once closed */
	bool refused = false;                       /* submit returned an error: every burst of the batch is delivered with -EIO */
	uint64_t seq = 0;                           /* position in submission order, taken when a completion thread picks the batch up */
	uint32_t epoch = 1;                         /* bumped per reuse: what Slot::ready must equal */
	const int16_t **src = nullptr;              /* by_reference: the slot's pointer array (trxhip_hostpipe_slot_sources) */
	std::vector<Slot> slot;
};
}  // namespace

struct BurstGatherer::Impl {
	BurstGathererConfig cfg;
	/* one entry per device of the list (one entry, the sigProcLibSetup() context, without a list) */
	struct Dev {
		trxhip_ctx *ctx = nullptr;
		bool own_ctx = false;                   /* created by this gatherer (trxsigproc_create_context): destroyed with it */
		trxhip_hostpipe *pipe = nullptr;
		std::atomic<uint64_t> n_batches{0};
	};
	std::vector<Dev> dev;
	std::vector<std::pair<const void *, size_t>> ranges;   /* registerBuffer(): applied to every pipe at start() */
	std::mutex life_mu;                         /* start() / stop() against each other */
	/* threads inside pushSlot() / pull(), counted on 64 cache lines: a consumer counts itself on its channel's line, a producer
	 * on its first channel's -- one shared word here was two contended read-modify-writes per call for every producer and
	 * consumer of the process, and bounded the whole stage at ~ 10 M calls/s (round 5, profiles/r05_gather.txt) */
	struct alignas(64) UserLine { std::atomic<int> n{0}; };
	UserLine users[64];
	std::atomic<bool> reconfig{false};          /* start() is replacing pipes, batches and rings: entrants leave at once */
	struct User {                               /* entry ticket of pushSlot() / pull() */
		std::atomic<int> &n;
		bool ok;
		User(Impl &m, size_t line) : n(m.users[line & 63].n)
		{
			n.fetch_add(1, std::memory_order_seq_cst);
			ok = !m.reconfig.load(std::memory_order_seq_cst);
		}
		~User() { n.fetch_sub(1, std::memory_order_seq_cst); }
	};
	trxhip_hostpipe *pipe_of(int b) const { return dev[(size_t)b % dev.size()].pipe; }
	int slot_of(int b) const { return b / (int)dev.size(); }
	void release_devices()
	{
		/* in reverse order of creation: the first pipe pinned the registered ranges, the others share its pin (include/trxhip.h) */
		for (size_t k = dev.size(); k-- > 0;) {
			Dev &d = dev[k];
			if (d.pipe) trxhip_hostpipe_destroy(d.pipe);
			if (d.own_ctx && d.ctx) trxsigproc_destroy_context(d.ctx);
		}
		dev.clear();
	}
	size_t payload = 0, entry_bytes = 0, stride = 0;
	std::vector<Batch> batch;
	std::vector<Chan> chan;
	std::mutex mu;                              /* queues and batch roll-over; NOT taken on the per-burst path */
	std::condition_variable cv_work, cv_space, cv_done;
	alignas(64) std::atomic<int> filling{-1};   /* read by every push, written at roll-over only */
	alignas(64) int pad_ = 0;
	std::deque<int> free_q, closed_q, flight_q;
	std::atomic<bool> stopping{false}, running{false};
	bool submitter_done = false;                /* under mu: nothing more will enter flight_q */
	std::thread submitter;
	std::vector<std::thread> completers;        /* one per device entry (or cfg.n_completers): see complete_loop() */
	size_t n_compl_active = 0;                  /* set by start() before the completion threads exist */
	std::mutex deliver_mu;                      /* sleepers of the delivery order (several completers only) */
	std::condition_variable cv_deliver;
	/* hand channel `ch`'s ring to the batch with sequence number `next`; wakes a completer that sleeps for its turn (the empty
	 * critical section orders the store against a sleeper's predicate check: no lost wake-up) */
	void pass_turn(Chan &ch, uint64_t next)
	{
		ch.deliver_seq.store(next, std::memory_order_release);
		if (n_compl_active > 1) {
			{ std::lock_guard<std::mutex> lk(deliver_mu); }
			cv_deliver.notify_all();
		}
	}
	uint64_t next_seq = 0;                      /* under mu */
	std::atomic<uint64_t> n_batches{0}, n_dropped{0}, n_rejected{0};

	/* make parked batch s the open one (caller holds mu, filling < 0 or being replaced): the ONLY place a batch is re-armed */
	void publish_locked(int s)
	{
		Batch &b = batch[s];
		b.count = 0;
		b.epoch++;
		b.first_ns.store(0, std::memory_order_relaxed);
		b.reserved.store(0, std::memory_order_release);
		filling.store(s, std::memory_order_release);
		cv_space.notify_all();
	}

	/* close batch f (full, or its first burst timed out) and make the next free one the filling batch; idempotent */
	void close_locked(int f)
	{
		if (filling.load(std::memory_order_relaxed) != f)
			return;
		Batch &b = batch[f];
		const uint64_t old = b.reserved.exchange(CLOSED, std::memory_order_acq_rel);   /* < CLOSED: f was the open batch */
		b.count = old < cfg.max_batch ? (uint32_t)old : (uint32_t)cfg.max_batch;
		closed_q.push_back(f);
		if (!free_q.empty()) {
			const int next = free_q.front();
			free_q.pop_front();
			publish_locked(next);
		} else {
			filling.store(-1, std::memory_order_release);
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
				for (uint32_t i = 0; i < b.count; i++)                       /* a producer may still be inside its memcpy */
					while (b.slot[i].ready.load(std::memory_order_acquire) != b.epoch)
						std::this_thread::yield();
				/* a failure surfaces in wait() -- except a refused by-reference batch (a burst outside the registered ranges:
				 * nothing was enqueued), which is remembered here */
				b.refused = (cfg.by_reference ? trxhip_hostpipe_submit_by_ref(pipe_of(s), slot_of(s), b.count)
							      : trxhip_hostpipe_submit(pipe_of(s), slot_of(s), b.count)) != TRXHIP_OK;
				dev[(size_t)s % dev.size()].n_batches++;
				lk.lock();
				n_batches++;
				flight_q.push_back(s);
				cv_done.notify_all();
				continue;
			}
			const int f = filling.load(std::memory_order_acquire);
			const int64_t first = f >= 0 ? batch[f].first_ns.load(std::memory_order_acquire) : 0;
			if (f >= 0 && first != 0) {
				const int64_t deadline = first + (int64_t)cfg.timeout_us * 1000;
				const int64_t now = now_ns();
				if (now >= deadline) {
					close_locked(f);
					continue;
				}
				timed_wait(cv_work, lk, std::chrono::nanoseconds(deadline - now));
			} else {
				/* nothing gathered: a push of a first burst does not take `mu`, so poll at the timeout's granularity */
				timed_wait(cv_work, lk, std::chrono::microseconds(cfg.timeout_us ? cfg.timeout_us : 1));
			}
		}
	}

	uint8_t *entry(Chan &ch, size_t k) { return ch.ring.data() + (k % cfg.fifo_depth) * entry_bytes; }

	void complete_loop()
	{
		std::vector<uint32_t> order, first, fill;
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv_done.wait(lk, [&] { return (stopping && submitter_done) || !flight_q.empty(); });
			if (flight_q.empty())
				return;                                        /* stopping, submitter gone, everything submitted delivered */
			const int s = flight_q.front();
			flight_q.pop_front();
			Batch &b = batch[s];
			b.seq = next_seq++;                                    /* flight_q is in submission order: so are the tickets */
			const uint64_t seq = b.seq;
			lk.unlock();
			const bool ok = trxhip_hostpipe_wait(pipe_of(s), slot_of(s)) == TRXHIP_OK && !b.refused;
			/* counting sort of the batch by channel (stable: arrival order within a channel is kept), then one lock and
			 * one wake-up per channel */
			const size_t nch = chan.size();
			order.resize(b.count);
			first.assign(nch + 1, 0);
			for (uint32_t i = 0; i < b.count; i++)
				first[b.slot[i].r.chan + 1]++;
			for (size_t c = 0; c < nch; c++)
				first[c + 1] += first[c];
			fill = first;
			for (uint32_t i = 0; i < b.count; i++)
				order[fill[b.slot[i].r.chan]++] = i;
			for (size_t c = 0; c < nch; c++) {
				Chan &ch = chan[c];
				/* Several completion threads: each has waited for its own batch and sorted it on its own; the rings are
				 * written in ticket order, channel by channel -- the thread of batch k + 1 follows one channel behind the
				 * thread of batch k.  (One thread: the ticket is always the ring's turn.) */
				if (ch.deliver_seq.load(std::memory_order_acquire) != seq) {
					/* a short spin (the thread ahead is usually one channel away), then SLEEP: it can be a whole GPU batch
					 * ahead when devices finish out of order, and the producers need the CPU */
					unsigned spin = 0;
					while (ch.deliver_seq.load(std::memory_order_acquire) != seq && ++spin < 256)
						;
					if (ch.deliver_seq.load(std::memory_order_acquire) != seq) {
						std::unique_lock<std::mutex> lk(deliver_mu);
						cv_deliver.wait(lk, [&] { return ch.deliver_seq.load(std::memory_order_acquire) == seq; });
					}
				}
				if (first[c] == first[c + 1]) {
					pass_turn(ch, seq + 1);
					continue;
				}
				size_t tail = ch.tail.load(std::memory_order_relaxed);
				{
					for (uint32_t k = first[c]; k < first[c + 1]; k++) {
						const uint32_t i = order[k];
						const Route &r = b.slot[i].r;
						/* cannot overflow: tail - head <= outstanding <= fifo_depth = ring size */
						uint8_t *e = entry(ch, tail++);
						Entry hd;
						if (ok) hd.res = b.h.results[i]; else memset(&hd.res, 0, sizeof(hd.res));
						hd.route = r;
						hd.code = !ok ? -EIO : (r.type == OFF ? -ENOENT : 0);
						hd.pkt_len = (ok && b.h.pkt) ? b.h.pkt_len[i] : 0;
						memcpy(e, &hd, sizeof(hd));
						if (ok && b.h.pkt)
							memcpy(e + sizeof(Entry), b.h.pkt + (size_t)i * stride, hd.pkt_len);
						else if (ok && b.h.soft && !hd.res.idle) {
							/* the kernels report nbits = 444 for an 8-PSK detection whatever the row width: never copy
							 * more than the row (and the ring entry) holds */
							const size_t nb = 4u * hd.res.nbits_div4;
							memcpy(e + sizeof(Entry), b.h.soft + (size_t)i * stride, (nb < stride ? nb : stride) * sizeof(float));
						}
					}
				}
				ch.tail.store(tail, std::memory_order_seq_cst);          /* publish; then look for a sleeping consumer */
				pass_turn(ch, seq + 1);
				if (ch.waiting.load(std::memory_order_seq_cst)) {
					std::lock_guard<std::mutex> g(ch.mu);
					ch.cv.notify_one();
				}
			}
			lk.lock();
			/* parked batches stay CLOSED (a stale reservation must fail); re-armed only when published */
			if (filling.load(std::memory_order_relaxed) < 0 && !stopping)
				publish_locked(s);
			else
				free_q.push_back(s);
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
	impl_->release_devices();
	delete impl_;
}

/* the device list: cfg.devices[], else TRXHIP_DEVICES=0,1,..., else empty (= the sigProcLibSetup() context alone) */
static std::vector<int> device_list(const BurstGathererConfig &cfg)
{
	std::vector<int> v;
	if (cfg.n_devices > 0) {
		for (int k = 0; k < cfg.n_devices && k < TRX_GATHERER_MAX_DEVICES; k++)
			v.push_back(cfg.devices[k]);
	} else if (const char *e = getenv("TRXHIP_DEVICES")) {
		while (*e && v.size() < TRX_GATHERER_MAX_DEVICES) {
			char *end = nullptr;
			const long d = strtol(e, &end, 10);
			if (end == e)
				break;
			v.push_back((int)d);
			e = (*end == ',') ? end + 1 : end;
		}
	}
	return v;
}

bool BurstGatherer::start()
{
	Impl &m = *impl_;
	std::lock_guard<std::mutex> life(m.life_mu);
	if (m.running || !trxsigproc_context() || m.cfg.chans == 0 || m.cfg.chans > 65535)
		return false;
	trxhip_hostpipe_cfg c;
	memset(&c, 0, sizeof(c));
	c.max_bursts = (uint32_t)m.cfg.max_batch;
	c.depth = m.cfg.depth;
	c.burst_len = (int32_t)m.cfg.burst_len;
	c.sps = m.cfg.sps;
	size_t stride, payload;
	if (m.cfg.trxd_version < 0) {
		c.soft_stride = m.cfg.egprs ? EDGE_BURST_NBITS : NORMAL_BURST_NBITS;
		stride = c.soft_stride;
		payload = stride * sizeof(float);
	} else {
		if (m.cfg.trxd_version > 1)
			return false;
		c.pkt_stride = m.cfg.egprs ? 456 : 160;
		stride = c.pkt_stride;
		payload = stride;
	}
	c.flags = TRXHIP_FLAG_SLICE | (m.cfg.exact_demod ? TRXHIP_FLAG_EXACT_DEMOD : 0) | (m.cfg.use_va ? TRXHIP_FLAG_USE_VA : 0);
	c.threshold = BURST_THRESH;
	c.full_scale = (float)m.cfg.rxFullScale;
	c.rssi_offset = (float)m.cfg.rssi_offset;
	if (m.cfg.n_paths < 0 || m.cfg.n_paths > 8 || m.cfg.n_devices < 0 || m.cfg.n_devices > TRX_GATHERER_MAX_DEVICES)
		return false;
	c.n_paths = m.cfg.n_paths > 1 ? m.cfg.n_paths : 0;

	/* From here on the previous run's pinned slots, batches and rings are replaced: no push() / pull() may be inside.  They
	 * count themselves in `users`; with `reconfig` raised new entrants leave at once, the ones already inside finish (a
	 * stopped gatherer blocks nobody: pull() on an empty FIFO returns -EIO, pushSlot() returns 0). */
	m.reconfig.store(true, std::memory_order_seq_cst);
	for (const Impl::UserLine &u : m.users)
		while (u.n.load(std::memory_order_seq_cst) != 0)
			std::this_thread::yield();
	struct Lower { std::atomic<bool> &f; ~Lower() { f.store(false, std::memory_order_seq_cst); } } lower{m.reconfig};

	m.release_devices();                                           /* restart: the previous run's contexts, staging slots, streams */
	const std::vector<int> list = device_list(m.cfg);
	m.dev = std::vector<Impl::Dev>(list.empty() ? 1 : list.size());
	for (size_t k = 0; k < m.dev.size(); k++) {
		Impl::Dev &d = m.dev[k];
		if (list.empty()) {
			d.ctx = trxsigproc_context();
		} else {
			d.ctx = trxsigproc_create_context(list[k]);
			d.own_ctx = true;
		}
		if (!d.ctx || trxhip_hostpipe_create(d.ctx, &c, &d.pipe) != TRXHIP_OK) {
			m.release_devices();
			return false;
		}
		if (m.cfg.by_reference)
			for (const auto &r : m.ranges)
				if (trxhip_hostpipe_register_host(d.pipe, r.first, r.second) != TRXHIP_OK) {
					m.release_devices();
					return false;
				}
	}
	m.stride = stride;
	m.payload = payload;
	m.entry_bytes = (sizeof(Entry) + m.payload + 15) & ~(size_t)15;
	const int n_batch = m.cfg.depth * (int)m.dev.size();
	std::lock_guard<std::mutex> g(m.mu);
	m.batch = std::vector<Batch>(n_batch);
	m.free_q.clear(); m.closed_q.clear(); m.flight_q.clear();
	for (int s = 0; s < n_batch; s++) {
		trxhip_hostpipe_slot_buffers(m.pipe_of(s), m.slot_of(s), &m.batch[s].h);
		if (m.cfg.by_reference && trxhip_hostpipe_slot_sources(m.pipe_of(s), m.slot_of(s), &m.batch[s].src) != TRXHIP_OK) {
			m.batch.clear();
			m.release_devices();                                   /* (release_devices() takes no lock: contexts and pipes only) */
			return false;
		}
		m.batch[s].slot = std::vector<Slot>(m.cfg.max_batch);
		for (size_t k = 0; k < m.cfg.max_batch; k++)
			m.batch[s].slot[k].ready.store(0, std::memory_order_relaxed);
		if (s)
			m.free_q.push_back(s);                                 /* opened in the order 0, 1, 2, ...: round-robin over the devices */
	}
	if (m.chan.size() != m.cfg.chans) {
		m.chan = std::vector<Chan>(m.cfg.chans);
		for (size_t c2 = 0; c2 < m.cfg.chans; c2++)
			m.chan[c2].trxd_version.store(m.cfg.trxd_version < 0 ? 0 : m.cfg.trxd_version, std::memory_order_relaxed);
	}
	for (size_t c2 = 0; c2 < m.cfg.chans; c2++) {                  /* empty FIFOs (a stopped run may have left bursts behind) */
		Chan &ch = m.chan[c2];
		ch.ring.assign(m.cfg.fifo_depth * m.entry_bytes, 0);
		ch.head = 0;
		ch.tail.store(0, std::memory_order_relaxed);
		ch.deliver_seq.store(0, std::memory_order_relaxed);
		ch.outstanding.store(0, std::memory_order_relaxed);
	}
	m.stopping = false;
	m.submitter_done = false;
	m.publish_locked(0);
	m.running = true;
	m.submitter = std::thread([&m] { m.submit_loop(); });
	m.next_seq = 0;
	size_t n_compl = m.cfg.n_completers > 0 ? (size_t)m.cfg.n_completers : m.dev.size();
	if (const char *e = getenv("TRXHIP_COMPLETERS")) if (atoi(e) > 0) n_compl = (size_t)atoi(e);
	if (n_compl > 16) n_compl = 16;
	m.n_compl_active = n_compl;
	m.completers.clear();
	for (size_t k = 0; k < n_compl; k++)
		m.completers.emplace_back([&m] { m.complete_loop(); });
	return true;
}

void BurstGatherer::stop()
{
	Impl &m = *impl_;
	std::lock_guard<std::mutex> life(m.life_mu);                   /* two stop() calls: the second finds it stopped */
	if (!m.running)
		return;
	{
		std::lock_guard<std::mutex> g(m.mu);
		m.stopping = true;
		m.cv_work.notify_all();
		m.cv_done.notify_all();
		m.cv_space.notify_all();
	}
	m.submitter.join();                                            /* it may have been inside trxhip_hostpipe_submit() */
	{
		std::lock_guard<std::mutex> g(m.mu);
		m.submitter_done = true;
		m.cv_done.notify_all();
	}
	for (std::thread &t : m.completers)
		t.join();                                                  /* they deliver every batch that was submitted */
	m.completers.clear();
	{
		/* gathered but never submitted: discarded (pull() on an empty FIFO of a stopped gatherer returns -EIO) */
		std::lock_guard<std::mutex> g(m.mu);
		const int f = m.filling.load(std::memory_order_relaxed);
		if (f >= 0)
			m.batch[f].reserved.store(CLOSED, std::memory_order_release);
		m.filling.store(-1, std::memory_order_release);
		m.closed_q.clear();
	}
	for (size_t c = 0; c < m.chan.size(); c++) {
		std::lock_guard<std::mutex> g(m.chan[c].mu);
		m.chan[c].cv.notify_all();
	}
	m.running = false;
}

bool BurstGatherer::push(size_t c, const BurstRequest &rq)
{
	return pushSlot(&c, &rq, 1, NULL) == 1;
}

size_t BurstGatherer::pushSlot(const size_t *chans, const BurstRequest *rqs, size_t n, bool *accepted)
{
	Impl &m = *impl_;
	Impl::User user(m, (chans && n) ? chans[0] : 0);
	if (!user.ok || !chans || !rqs || n > 65535 || m.stopping || !m.running) {     /* (ok_idx[] holds 16-bit positions) */
		if (accepted)
			for (size_t k = 0; k < n; k++) accepted[k] = false;
		return 0;
	}
	/* the FIFO rule first, per channel (radioInterface.cpp:277-280: FIFO full, burst deleted) */
	enum { STACK = 64 };
	uint16_t ok_idx_stack[STACK];
	std::vector<uint16_t> ok_idx_heap;
	uint16_t *ok_idx = ok_idx_stack;
	if (n > STACK) {
		ok_idx_heap.resize(n);
		ok_idx = ok_idx_heap.data();
	}
	size_t n_ok = 0;
	for (size_t k = 0; k < n; k++) {
		bool ok = chans[k] < m.chan.size() && rqs[k].iq;
		if (ok && rqs[k].type == EDGE && !m.cfg.egprs) {              /* 444-bit rows were not configured (cfg->egprs) */
			m.n_rejected++;
			ok = false;
		}
		if (ok) {
			Chan &ch = m.chan[chans[k]];
			if (ch.outstanding.fetch_add(1, std::memory_order_acq_rel) >= m.cfg.fifo_depth) {
				ch.outstanding.fetch_sub(1, std::memory_order_relaxed);
				m.n_dropped++;
				ok = false;
			}
		}
		if (accepted)
			accepted[k] = ok;
		if (ok)
			ok_idx[n_ok++] = (uint16_t)k;
	}
	/* then one reservation of consecutive staging slots for all of them (two when the batch rolls over in between) */
	size_t done = 0;
	while (done < n_ok) {
		const int f = m.filling.load(std::memory_order_acquire);
		if (f < 0) {
			/* all staging batches closed or in flight: only possible when depth * max_batch < chans * fifo_depth */
			std::unique_lock<std::mutex> lk(m.mu);
			m.cv_space.wait(lk, [&] { return m.stopping || m.filling.load(std::memory_order_acquire) >= 0; });
			if (m.stopping) {
				for (size_t k = done; k < n_ok; k++) {
					m.chan[chans[ok_idx[k]]].outstanding.fetch_sub(1, std::memory_order_relaxed);
					if (accepted) accepted[ok_idx[k]] = false;
				}
				return done;
			}
			continue;
		}
		Batch &b = m.batch[f];
#ifdef TRX_GATHERER_TEST_STALL
		TRX_GATHERER_TEST_STALL();      /* tests/gatherer_stub: deschedule here for longer than a batch round trip */
#endif
		const uint32_t want = (uint32_t)(n_ok - done);
		const uint64_t idx0_ = b.reserved.fetch_add(want, std::memory_order_acq_rel);  /* the only shared write of a push */
		if (idx0_ >= m.cfg.max_batch) {
			/* full (its last taker is rolling it over), or closed / in flight / parked (this thread read `filling` a
			 * while ago): nothing was reserved -- the word is >= max_batch and stays so until the batch is re-opened */
			if (m.filling.load(std::memory_order_acquire) == f)
				std::this_thread::yield();
			continue;
		}
		const uint32_t idx0 = (uint32_t)idx0_;
		const uint32_t got = (idx0 + want <= m.cfg.max_batch) ? want : (uint32_t)m.cfg.max_batch - idx0;
		if (idx0 == 0) {
			b.first_ns.store(now_ns(), std::memory_order_release);     /* arms the timeout */
			m.cv_work.notify_one();
		}
		for (uint32_t j = 0; j < got; j++) {
			const size_t k = ok_idx[done + j];
			const BurstRequest &rq = rqs[k];
			const uint32_t idx = idx0 + j;
			Route &r = b.slot[idx].r;
			r.chan = (uint16_t)chans[k];
			r.type = (uint8_t)rq.type;
			r.tn = rq.tn;
			r.fn = rq.fn;
			if (m.cfg.by_reference) {
				b.src[idx] = rq.iq;                                     /* the device fetches it from the registered ring */
			} else {
				/* the burst goes straight into the pinned slot the DMA engine reads from */
				const size_t burst_i16 = m.cfg.burst_len * 2 * (m.cfg.n_paths > 1 ? (size_t)m.cfg.n_paths : 1);
				memcpy(b.h.iq + (size_t)idx * burst_i16, rq.iq, burst_i16 * sizeof(int16_t));
			}
			trxhip_burst_params &p = b.h.params[idx];
			p.type = (uint8_t)rq.type;
			p.tsc = (uint8_t)rq.tsc;
			p.max_toa = (uint16_t)rq.max_toa;
			p.reserved = 0;
			if (b.h.meta) {
				trxhip_trxd_meta &mt = b.h.meta[idx];
				mt.fn = rq.fn;
				mt.tn = rq.tn;
				mt.version = (uint8_t)m.chan[chans[k]].trxd_version.load(std::memory_order_relaxed);
				mt.tss = 0;
				mt.reserved = 0;
			}
			b.slot[idx].ready.store(b.epoch, std::memory_order_release);
		}
		done += got;
		if (idx0 + want >= m.cfg.max_batch) {                      /* took the last slot: roll the batch over */
			std::lock_guard<std::mutex> g(m.mu);
			m.close_locked(f);
		}
	}
	return n_ok;
}

int BurstGatherer::pull(size_t c, BurstIndication *bi, uint8_t *pkt, size_t *pkt_len)
{
	Impl &m = *impl_;
	Impl::User user(m, c + 32);                                        /* (a channel's consumer and its producer on different lines) */
	if (!user.ok || c >= m.chan.size() || !bi)
		return -EIO;
	Chan &ch = m.chan[c];
	if (ch.tail.load(std::memory_order_acquire) == ch.head) {         /* empty: block, as the reference's FIFO read does (:683) */
		std::unique_lock<std::mutex> lk(ch.mu);
		ch.waiting.store(1, std::memory_order_seq_cst);
		ch.cv.wait(lk, [&] { return ch.tail.load(std::memory_order_seq_cst) != ch.head || m.stopping || !m.running ||
					    m.reconfig.load(std::memory_order_seq_cst); });
		ch.waiting.store(0, std::memory_order_relaxed);
		if (ch.tail.load(std::memory_order_acquire) == ch.head)
			return -EIO;
	}
	const uint8_t *e = m.entry(ch, ch.head);
	Entry hd;
	memcpy(&hd, e, sizeof(hd));
	BurstRequest rq;
	memset(&rq, 0, sizeof(rq));
	rq.type = (CorrType)hd.route.type;
	rq.fn = hd.route.fn;
	rq.tn = hd.route.tn;
	const bool floats = m.cfg.trxd_version < 0;
	trxsigproc_fill_indication(*bi, rq, hd.res, floats ? reinterpret_cast<const float *>(e + sizeof(Entry)) : NULL, m.stride,
				   m.cfg.rssi_offset);
	if (!floats && pkt)
		memcpy(pkt, e + sizeof(Entry), hd.pkt_len);
	if (pkt_len)
		*pkt_len = hd.pkt_len;
	ch.head++;
	ch.outstanding.fetch_sub(1, std::memory_order_release);
	return hd.code;
}

bool BurstGatherer::setTrxdVersion(size_t c, int version)
{
	Impl &m = *impl_;
	if (m.cfg.trxd_version < 0 || version < 0 || version > 1)
		return false;                                                  /* float mode is gatherer-wide */
	std::lock_guard<std::mutex> life(m.life_mu);                       /* (creates the channel table before the first start()) */
	if (m.chan.size() != m.cfg.chans) {                                /* before the first start(): create the channels now */
		m.chan = std::vector<Chan>(m.cfg.chans);
		for (size_t k = 0; k < m.cfg.chans; k++)
			m.chan[k].trxd_version.store(m.cfg.trxd_version, std::memory_order_relaxed);
	}
	if (c >= m.chan.size())
		return false;
	m.chan[c].trxd_version.store(version, std::memory_order_relaxed);
	return true;
}

bool BurstGatherer::registerBuffer(const void *base, size_t bytes)
{
	Impl &m = *impl_;
	std::lock_guard<std::mutex> life(m.life_mu);
	if (m.running || !m.cfg.by_reference || !base || bytes == 0 || m.ranges.size() >= 8)
		return false;
	m.ranges.emplace_back(base, bytes);
	return true;
}

uint64_t BurstGatherer::rejected() const { return impl_->n_rejected.load(); }
uint64_t BurstGatherer::batches() const { return impl_->n_batches.load(); }
uint64_t BurstGatherer::dropped() const { return impl_->n_dropped.load(); }
const BurstGathererConfig &BurstGatherer::config() const { return impl_->cfg; }
size_t BurstGatherer::devices() const { return impl_->dev.size(); }
uint64_t BurstGatherer::batchesOn(size_t entry) const { return entry < impl_->dev.size() ? impl_->dev[entry].n_batches.load() : 0; }

TRX_SHIM_NS_END
