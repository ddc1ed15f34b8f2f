// gatherer_stress.cpp -- BurstGatherer under ThreadSanitizer / AddressSanitizer, on the CPU stand-in of the hostpipe.
// One producer thread and one consumer thread per channel on an oversubscribed machine; checks the class's contract
// (radioInterface.cpp:272-291, Transceiver.cpp:1229-1253): every accepted burst is delivered exactly once, to its own
// channel, in push order; dropped bursts (FIFO full) are never delivered.
//   gatherer_stress <channels> <pushes_per_channel> <max_batch> <timeout_us> <trxd_version> [restart] [n_devices] [churn] [by_ref] [n_completers]
// by_ref = 1: BurstGathererConfig::by_reference -- the producers keep their bursts in a registered ring of fifo_depth + 1 burst
// periods per channel (the contract of trxBatch.h) and push addresses; under TSan this also checks that a ring position is only
// rewritten after the pull of its previous burst happened-before.
// n_devices > 0: the multi-device dispatcher on that many fake devices (every one must see its share of the batches, the
// per-channel order must not show which device ran a batch).  churn = 1: a third pass in which stop() / start() are called
// WHILE producers and consumers are inside push() / pull() (round 3's advisor finding: start() freed the pinned slots and
// rings under them) -- nothing may crash or trip the sanitizer; deliveries between two restarts stay ordered.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trxBatch.h"
#ifdef TRXHIP_SA_NS
using namespace trxhip_sa;
#endif

/* by_ref: channel c's ring, position k (k = bursts of the channel accepted so far, modulo the ring); one spare burst behind */
static bool g_by_ref = false;
static size_t g_ring_slots = 33;
static std::vector<int16_t> g_ring;
static int16_t *ring_burst(size_t c, uint64_t k) { return g_ring.data() + (c * g_ring_slots + (size_t)(k % g_ring_slots)) * 1250; }
static int16_t *spare_burst(size_t nch) { return g_ring.data() + nch * g_ring_slots * 1250; }

extern "C" int stub_live_pipes(void);
extern "C" int stub_live_contexts(void);
extern "C" long stub_submits_on(int device);

/* stop() / start() under fire: producers and consumers keep calling while the gatherer is cycled */
static void churn(BurstGatherer &g, size_t nch, std::atomic<long> &errors)
{
	std::atomic<bool> quit{false};
	std::atomic<uint64_t> pushed{0}, pulled{0};
	std::vector<std::thread> th;
	for (size_t c = 0; c < nch; c++) {
		th.emplace_back([&, c] {
			std::vector<int16_t> own(625 * 2, 0);
			uint32_t fn = 0;
			uint64_t acc = 0;
			while (!quit.load()) {
				fn++;
				int16_t *iq = g_by_ref ? ring_burst(c, acc) : own.data();
				iq[0] = (int16_t)(fn & 0x7fff); iq[1] = (int16_t)(fn >> 15); iq[2] = (int16_t)c;
				BurstRequest rq;
				memset(&rq, 0, sizeof(rq));
				rq.iq = iq; rq.type = TSC; rq.tsc = fn & 7; rq.max_toa = 3; rq.fn = fn; rq.tn = fn & 7;
				if (g.push(c, rq)) { pushed++; acc++; } else std::this_thread::yield();
			}
		});
		th.emplace_back([&, c] {
			BurstIndication bi;
			uint32_t last_fn = 0;
			while (!quit.load()) {
				memset(&bi, 0, sizeof(bi));
				const int rc = g.pull(c, &bi);
				if (rc == -5) { last_fn = 0; std::this_thread::yield(); continue; }   /* stopped: a restart begins from empty FIFOs */
				pulled++;
				if ((size_t)bi.energy != c) { errors++; fprintf(stderr, "churn chan %zu: foreign burst (chan %g)\n", c, bi.energy); }
				if (bi.fn <= last_fn) {
					/* a restart between two pulls resets the FIFO: fn may go back only across a -EIO ... or when the consumer did
					 * not see the stop at all (it was not pulling): accept a drop back, never a repeat of the same fn */
					if (bi.fn == last_fn) { errors++; fprintf(stderr, "churn chan %zu: fn %u twice\n", c, bi.fn); }
				}
				last_fn = bi.fn;
			}
		});
	}
	for (int k = 0; k < 40; k++) {
		std::this_thread::sleep_for(std::chrono::milliseconds(5));
		g.stop();
		if (k & 1) g.stop();                                           /* a second stop() finds it stopped */
		if (!g.start()) { errors++; fprintf(stderr, "churn: restart %d failed\n", k); break; }
	}
	quit.store(true);
	g.stop();                                                          /* wakes blocked consumers; producers see `stopping` */
	for (auto &t : th) t.join();
	printf("churn pushed %llu pulled %llu\n", (unsigned long long)pushed.load(), (unsigned long long)pulled.load());
	if (pulled.load() == 0 || pulled.load() > pushed.load()) { errors++; fprintf(stderr, "churn accounting\n"); }
}
static const uint32_t END_FN = 0xfffff0;       /* below 2^24: exact in the stub's float echo */

static int run_once(BurstGatherer &g, size_t nch, uint32_t per_chan, int version, std::atomic<long> &errors)
{
	std::vector<std::atomic<uint32_t>> accepted(nch);
	std::vector<std::thread> th;
	std::atomic<size_t> producers_left{nch};
	std::atomic<uint64_t> refused{0};                              /* pushes that returned false (regular bursts and end-marker retries) */
	const uint64_t dropped0 = g.dropped();
	for (size_t c = 0; c < nch; c++)
		accepted[c].store(0);
	for (size_t c = 0; c < nch; c++) {
		th.emplace_back([&, c] {                                   /* producer of channel c */
			std::vector<int16_t> own(625 * 2, 0);
			uint32_t acc = 0;
			for (uint32_t fn = 1; fn <= per_chan; fn++) {
				int16_t *iq = g_by_ref ? ring_burst(c, acc) : own.data();
				iq[0] = (int16_t)(fn & 0x7fff); iq[1] = (int16_t)(fn >> 15); iq[2] = (int16_t)c;
				BurstRequest rq;
				memset(&rq, 0, sizeof(rq));
				rq.iq = iq; rq.type = (fn % 97 == 0) ? OFF : TSC; rq.tsc = fn & 7; rq.max_toa = 3; rq.fn = fn; rq.tn = fn & 7;
				/* seven bursts in eight wait for room (so that most are accepted and the ordering check has something to
				 * look at); the eighth is dropped when the FIFO is full, as the reference's producer does */
				bool ok = g.push(c, rq);
				while (!ok && (fn & 7) != 0) { refused++; std::this_thread::yield(); ok = g.push(c, rq); }
				if (ok) acc++; else refused++;
				if ((fn & 1023) == 0) std::this_thread::yield();
			}
			/* end marker (retried until the FIFO takes it): tells the consumer that nothing follows */
			int16_t *iq = g_by_ref ? ring_burst(c, acc) : own.data();
			iq[0] = (int16_t)(END_FN & 0x7fff); iq[1] = (int16_t)(END_FN >> 15); iq[2] = (int16_t)c;
			BurstRequest rq;
			memset(&rq, 0, sizeof(rq));
			rq.iq = iq; rq.type = TSC; rq.fn = END_FN;
			while (!g.push(c, rq)) { refused++; std::this_thread::yield(); }
			accepted[c].store(acc, std::memory_order_release);
			producers_left--;
		});
		th.emplace_back([&, c] {                                   /* consumer of channel c */
			BurstIndication bi;
			uint8_t pkt[TRXD_MAX_PKT_LEN];
			uint32_t last_fn = 0, got = 0;
			for (;;) {
				size_t plen = 0;
				memset(&bi, 0, sizeof(bi));
				const int rc = g.pull(c, &bi, version >= 0 ? pkt : NULL, &plen);
				if (rc == -5) { errors++; fprintf(stderr, "chan %zu: pull -EIO\n", c); break; }
				if (bi.fn == END_FN) break;
				got++;
				if (bi.fn <= last_fn) { errors++; fprintf(stderr, "chan %zu: fn %u after %u\n", c, bi.fn, last_fn); }
				last_fn = bi.fn;
				const bool off = (bi.fn % 97 == 0);
				if ((rc == -2) != off) { errors++; fprintf(stderr, "chan %zu fn %u: rc %d\n", c, bi.fn, rc); }
				if (!off) {
					if ((uint32_t)bi.toa != bi.fn || (size_t)bi.energy != c) { errors++; fprintf(stderr, "chan %zu fn %u: echoed fn %u chan %g\n", c, bi.fn, (uint32_t)bi.toa, bi.energy); }
					if (version >= 0) {
						const uint32_t pfn = ((uint32_t)pkt[1] << 24) | ((uint32_t)pkt[2] << 16) | ((uint32_t)pkt[3] << 8) | pkt[4];
						if (pfn != bi.fn || (pkt[0] >> 4) != (int)(c & 1)) { errors++; fprintf(stderr, "chan %zu fn %u: packet fn %u version %d\n", c, bi.fn, pfn, pkt[0] >> 4); }
					} else if (bi.nbits != 148 || bi.rx_burst[0] != (float)(bi.fn & 1)) { errors++; fprintf(stderr, "chan %zu fn %u: soft row\n", c, bi.fn); }
				}
			}
			while (producers_left.load() != 0) std::this_thread::yield();      /* accepted[] is final */
			if (got != accepted[c].load()) { errors++; fprintf(stderr, "chan %zu: got %u of %u accepted\n", c, got, accepted[c].load()); }
		});
	}
	for (auto &t : th) t.join();
	uint64_t acc = 0;
	for (size_t c = 0; c < nch; c++) acc += accepted[c].load();
	printf("accepted %llu of %llu, dropped %llu, batches %llu\n", (unsigned long long)acc, (unsigned long long)nch * per_chan,
	       (unsigned long long)g.dropped(), (unsigned long long)g.batches());
	if (g.dropped() - dropped0 != refused.load() || acc + refused.load() < (uint64_t)nch * per_chan) { errors++; fprintf(stderr, "drop accounting: gatherer %llu, producers %llu\n", (unsigned long long)(g.dropped() - dropped0), (unsigned long long)refused.load()); }
	return 0;
}

int main(int argc, char **argv)
{
	const size_t nch = argc > 1 ? atoi(argv[1]) : 16;
	const uint32_t per_chan = argc > 2 ? atoi(argv[2]) : 65536;
	BurstGathererConfig cfg;
	memset(&cfg, 0, sizeof(cfg));
	cfg.chans = nch; cfg.max_batch = argc > 3 ? atoi(argv[3]) : 64; cfg.timeout_us = argc > 4 ? atoi(argv[4]) : 50;
	cfg.fifo_depth = 32; cfg.sps = 4; cfg.burst_len = 625; cfg.rxFullScale = 32767.0; cfg.rssi_offset = 0.0;
	cfg.egprs = false; cfg.trxd_version = argc > 5 ? atoi(argv[5]) : -1; cfg.depth = 4;
	const bool restart = argc > 6 && atoi(argv[6]);
	const int n_dev = argc > 7 ? atoi(argv[7]) : 0;
	const bool do_churn = argc > 8 && atoi(argv[8]);
	cfg.n_devices = n_dev;
	for (int k = 0; k < n_dev; k++) cfg.devices[k] = k;
	cfg.n_completers = argc > 10 ? atoi(argv[10]) : 0;             /* several completion threads: per-channel order must not show it */
	g_by_ref = argc > 9 && atoi(argv[9]);
	cfg.by_reference = g_by_ref;
	g_ring_slots = cfg.fifo_depth + 1;
	g_ring.assign((nch * g_ring_slots + 1) * 1250, 0);
	std::atomic<long> errors{0};
	{
		BurstGatherer g(cfg);
		if (g_by_ref) {
			if (!g.registerBuffer(g_ring.data(), g_ring.size() * sizeof(int16_t))) { fprintf(stderr, "registerBuffer failed\n"); return 2; }
		} else if (g.registerBuffer(g_ring.data(), 16)) { errors++; fprintf(stderr, "registerBuffer accepted without by_reference\n"); }
		if (cfg.trxd_version >= 0)
			for (size_t c = 0; c < nch; c++)
				if (!g.setTrxdVersion(c, (int)(c & 1))) errors++;      /* per-channel header version (mVersionTRXD[chan]) */
		if (!g.start()) { fprintf(stderr, "start failed\n"); return 2; }
		run_once(g, nch, per_chan, cfg.trxd_version, errors);
		if (n_dev > 0) {
			/* every fake device got its share */
			if (g.devices() != (size_t)n_dev || stub_live_contexts() != n_dev || stub_live_pipes() != n_dev) { errors++; fprintf(stderr, "device entries\n"); }
			long lo = 1L << 60, hi = 0, sum = 0;
			for (int k = 0; k < n_dev; k++) {
				const long v = stub_submits_on(k);
				if ((uint64_t)v != g.batchesOn(k)) { errors++; fprintf(stderr, "device %d: stub saw %ld, gatherer %llu\n", k, v, (unsigned long long)g.batchesOn(k)); }
				lo = v < lo ? v : lo; hi = v > hi ? v : hi; sum += v;
			}
			printf("devices %d batches_per_device %ld..%ld\n", n_dev, lo, hi);
			/* (one completion thread per device: a staging batch is re-opened when ITS device has finished, so the entries
			 * take turns only as long as they keep pace -- every one must still carry its share) */
			if ((uint64_t)sum != g.batches() || hi - lo > 2 + sum / 50) { errors++; fprintf(stderr, "round-robin: %ld..%ld of %llu\n", lo, hi, (unsigned long long)g.batches()); }
		}
		/* an EDGE slot on a gatherer without egprs rows is refused, not written past the 148-float payload */
		{
			BurstRequest rq; memset(&rq, 0, sizeof(rq));
			rq.iq = spare_burst(nch); rq.type = EDGE; rq.fn = 1;
			if (g.push(0, rq) || g.rejected() != 1) { errors++; fprintf(stderr, "EDGE push on !egprs accepted\n"); }
			if (g_by_ref) {
				/* a burst outside every registered range: accepted by push() (it only records the address), its batch is refused
				 * by the pipe and comes back as -EIO; the gatherer keeps running */
				static int16_t outside[1250];
				BurstIndication bi;
				rq.iq = outside; rq.type = TSC; rq.fn = 3;
				if (!g.push(0, rq) || g.pull(0, &bi) != -5 || bi.fn != 3) { errors++; fprintf(stderr, "unregistered address not refused\n"); }
				rq.iq = spare_burst(nch); rq.fn = 4;
				memset(spare_burst(nch), 0, 2500);
				if (!g.push(0, rq) || g.pull(0, &bi) != 0 || bi.fn != 4) { errors++; fprintf(stderr, "push after a refused batch\n"); }
			}
		}
		if (restart) {
			/* stop with bursts gathered but not submitted, restart: one pipe alive, FIFOs empty, full run again */
			BurstRequest rq; memset(&rq, 0, sizeof(rq));
			rq.iq = spare_burst(nch); rq.type = TSC; rq.fn = 7;
			for (int k = 0; k < 5; k++) g.push(0, rq);
			g.stop();
			BurstIndication bi;
			int n_eio = 0, n_ok = 0;
			for (int k = 0; k < 6; k++) { const int rc = g.pull(0, &bi); if (rc == -5) { n_eio++; break; } n_ok++; }
			if (n_eio != 1) { errors++; fprintf(stderr, "pull after stop: %d delivered, no -EIO\n", n_ok); }
			if (!g.start()) { errors++; fprintf(stderr, "restart failed\n"); }
			if (stub_live_pipes() != (n_dev > 0 ? n_dev : 1)) { errors++; fprintf(stderr, "%d hostpipes alive after restart\n", stub_live_pipes()); }
			run_once(g, nch, per_chan / 4 + 1, cfg.trxd_version, errors);
		}
		if (do_churn) {
			g.stop();
			if (!g.start()) { errors++; fprintf(stderr, "start before churn failed\n"); }
			churn(g, nch, errors);
		}
		g.stop();
	}
	if (stub_live_pipes() != 0 || stub_live_contexts() != 0) { errors++; fprintf(stderr, "%d hostpipes / %d contexts leaked\n", stub_live_pipes(), stub_live_contexts()); }
	/* EDGE rows with egprs = false used to overflow the ring entry: egprs gatherer, float mode, 444-bit rows delivered */
	{
		cfg.egprs = true; cfg.trxd_version = -1; cfg.chans = 1; cfg.by_reference = false;
		BurstGatherer g(cfg);
		if (!g.start()) return 2;
		std::vector<int16_t> iq(625 * 2, 0);
		iq[0] = 5;
		BurstRequest rq; memset(&rq, 0, sizeof(rq));
		rq.iq = iq.data(); rq.type = EDGE; rq.fn = 5;
		BurstIndication bi;
		if (!g.push(0, rq) || g.pull(0, &bi) != 0 || bi.nbits != 444 || bi.rx_burst[443] != (float)((5 + 443) & 1)) { errors++; fprintf(stderr, "EDGE row\n"); }
		g.stop();
	}
	printf("errors %ld\n", errors.load());
	return errors.load() ? 1 : 0;
}
