// sigproc_selftest.cpp -- exercises the sigProcLib-compatible host shim the way the reference's callers do
// (burst-gen.cpp:274-290 for the captured burst; Transceiver.cpp:768-803 for the detect -> demod -> slice
// sequence).  Reads inputs prepared by tests/test_gpu_host_shim.py, writes results as plain text/binary.
//   sigproc_selftest capture <cfile> <out.txt>
//   sigproc_selftest delay <cfile> <delay> <scale_re> <scale_im> <out.cf32>
//   sigproc_selftest batchva <iq.s16> <params.bin> <n> <rec.bin> <soft.bin>
//   sigproc_selftest va <cfile> <tsc> <out.f32>
//   sigproc_selftest sch <cfile> <0 full | 1 narrow | 2 buffer> <out.txt>
//   sigproc_selftest batch <iq.s16> <params.bin> <n> <sps> <burst_len> <out_results.bin> <out_soft.bin>
//   sigproc_selftest pullrv <iq.s16> <params.bin> <n> <chans> <muted_chan | -1> <exact 0|1> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include <atomic>
#include <chrono>
#include <thread>

#include "trxBatch.h"
#include "trxPullRadioVector.h"
#include "MultiArfcnRx.h"

// "signalvector is owning despite claiming not to" (Transceiver.cpp:648-654): a vector built over memory it must not
// delete[] gets a no-op deallocator, exactly as pullRadioVector() does for its shift buffer (:680)
static void dummy_free(void *) {}
static void *dummy_alloc(size_t) { return 0; }

// the layout every reference-compiled caller was built with (CommonLibs/Vector.h:72-76: three pointers + two function
// pointers; signalVector.h:48-51: two bools + enum)
static_assert(sizeof(signalVector) == 48, "signalVector layout");
static_assert(sizeof(SoftVector) == 40 && sizeof(complex) == 8, "SoftVector / complex layout");

extern "C" const char *trxsigproc_abi(void);
extern "C" void trxsigproc_abi_layout(size_t out[8]);

static std::vector<char> slurp(const char *path)
{
	std::vector<char> v;
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); exit(2); }
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	v.resize(n);
	if (fread(v.data(), 1, n, f) != (size_t)n) { perror("read"); exit(2); }
	fclose(f);
	return v;
}

// abi: what this executable's compiler sees vs what the library was compiled with (no GPU needed)
static int abi_report()
{
	size_t lib[8];
	trxsigproc_abi_layout(lib);
	const size_t mine[8] = { sizeof(signalVector), sizeof(SoftVector), sizeof(complex), sizeof(struct estim_burst_params),
				 offsetof(struct estim_burst_params, toa), offsetof(struct estim_burst_params, tsc),
				 offsetof(struct estim_burst_params, ci), sizeof(Vector<float>) };
	int bad = 0;
	printf("abi %s\n", trxsigproc_abi());
	{
		/* struct trx_ul_burst_ind as this executable's compiler lays it out vs the library's (trxPullRadioVector.h) */
		size_t lb[14];
		trxsigproc_abi_layout_bi(lb);
		const size_t mb[14] = { sizeof(struct trx_ul_burst_ind), offsetof(struct trx_ul_burst_ind, rx_burst),
					offsetof(struct trx_ul_burst_ind, nbits), offsetof(struct trx_ul_burst_ind, fn),
					offsetof(struct trx_ul_burst_ind, tn), offsetof(struct trx_ul_burst_ind, rssi),
					offsetof(struct trx_ul_burst_ind, toa), offsetof(struct trx_ul_burst_ind, noise),
					offsetof(struct trx_ul_burst_ind, idle), offsetof(struct trx_ul_burst_ind, modulation),
					offsetof(struct trx_ul_burst_ind, tss), offsetof(struct trx_ul_burst_ind, tsc),
					offsetof(struct trx_ul_burst_ind, ci), sizeof(enum Modulation) };
		static const char *const bn[14] = { "sizeof_trx_ul_burst_ind", "offsetof_bi_rx_burst", "offsetof_bi_nbits", "offsetof_bi_fn",
						    "offsetof_bi_tn", "offsetof_bi_rssi", "offsetof_bi_toa", "offsetof_bi_noise",
						    "offsetof_bi_idle", "offsetof_bi_modulation", "offsetof_bi_tss", "offsetof_bi_tsc",
						    "offsetof_bi_ci", "sizeof_enum_Modulation" };
		for (int k = 0; k < 14; k++) {
			printf("%s %zu %zu\n", bn[k], mb[k], lb[k]);
			bad |= mb[k] != lb[k];
		}
#ifdef TRX_HAVE_REFERENCE_PROTO_TRXD
		printf("trx_ul_burst_ind reference\n");
#else
		printf("trx_ul_burst_ind declared\n");
#endif
	}
	static const char *const name[8] = { "sizeof_signalVector", "sizeof_SoftVector", "sizeof_complex", "sizeof_ebp",
					     "offsetof_ebp_toa", "offsetof_ebp_tsc", "offsetof_ebp_ci", "sizeof_Vector_float" };
	for (int k = 0; k < 8; k++) {
		printf("%s %zu %zu\n", name[k], mine[k], lib[k]);
		bad |= mine[k] != lib[k];
	}
	/* the 5-argument constructor and head-room form pullRadioVector() and radioVector use */
	static complex buf[625];
	signalVector alias(buf, 0, 625, dummy_alloc, dummy_free);
	signalVector headroom(625, 41);
	printf("alias_size %zu alias_start %zu headroom_size %zu headroom_start %zu\n", alias.size(), alias.getStart(),
	       headroom.size(), headroom.getStart());
	struct estim_burst_params ebp;
	memset(&ebp, 0, sizeof(ebp));
	/* without a GPU (or before sigProcLibSetup) the calls fail the reference's way instead of touching the objects */
	printf("detect_without_setup %d\n", detectAnyBurst(alias, 0, BURST_THRESH, 4, TSC, 3, &ebp));
	SoftVector *sv = demodAnyBurst(alias, TSC, 4, &ebp);
	printf("demod_without_setup %s\n", sv ? "non-null" : "null");
	delete sv;
	return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	if (!strcmp(argv[1], "abi"))
		return abi_report();
	if (!sigProcLibSetup()) { fprintf(stderr, "no GPU\n"); return 3; }

	if (!strcmp(argv[1], "capture") && argc == 4) {
		std::vector<char> raw = slurp(argv[2]);
		size_t n = raw.size() / sizeof(complex);
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n, dummy_alloc, dummy_free);
		struct estim_burst_params ebp;
		int rc = detectAnyBurst(sv, 7, BURST_THRESH, 4, TSC, 40, &ebp);
		FILE *o = fopen(argv[3], "w");
		fprintf(o, "rc %d\ntoa %.9g\namp %.9g %.9g\nci %.9g\ntsc %u\n", rc, ebp.toa, ebp.amp.real(), ebp.amp.imag(), ebp.ci, ebp.tsc);
		if (rc > 0) {
			std::unique_ptr<SoftVector> soft(demodAnyBurst(sv, (CorrType)rc, 4, &ebp));
			fprintf(o, "nsoft %zu\nbits ", soft->size());
			for (size_t i = 0; i < 148; i++) fputc(soft->bit(i) ? '1' : '0', o);
			fputc('\n', o);
			/* second path: demod alone with caller-held parameters on a copy of the burst */
			signalVector copy(n);
			memcpy(copy.begin(), sv.begin(), sv.bytes());
			std::unique_ptr<SoftVector> soft2(demodAnyBurst(copy, (CorrType)rc, 4, &ebp));
			int same = soft2 && soft2->size() == soft->size() && !memcmp(soft2->begin(), soft->begin(), soft->bytes());
			fprintf(o, "demod_alone_identical %d\n", same);
			float sliced[148];
			vectorSlicer(sliced, soft->begin(), 148);
			fprintf(o, "sliced0 %.9g %.9g %.9g\n", sliced[0], sliced[73], sliced[147]);
		}
		fprintf(o, "energy %.9g\n", energyDetect(sv, 80));
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	if (!strcmp(argv[1], "delay") && argc == 7) {
		/* delayVector(burst, NULL, d) then scaleVector(*delay, s): the pair demodCommon() and ms_rx_lower.cpp:243-245 use */
		std::vector<char> raw = slurp(argv[2]);
		size_t n = raw.size() / sizeof(complex);
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n, dummy_alloc, dummy_free);
		std::unique_ptr<signalVector> d(delayVector(&sv, NULL, (float)atof(argv[3])));
		if (!d) return 4;
		scaleVector(*d, complex((float)atof(argv[4]), (float)atof(argv[5])));
		FILE *o = fopen(argv[6], "wb");
		fwrite(d->begin(), 1, d->bytes(), o);
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	if (!strcmp(argv[1], "va") && argc == 5) {
		/* Transceiver.cpp:782-784: scaleVector(*burst, 1/16383) then demodAnyBurst_va(*burst, TSC, 4, max_toa, tsc) */
		std::vector<char> raw = slurp(argv[2]);
		size_t n = raw.size() / sizeof(complex);
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n, dummy_alloc, dummy_free);
		scaleVector(sv, complex((float)(1. / (float)((1 << 14) - 1)), 0));
		std::unique_ptr<SoftVector> bits(demodAnyBurst_va(sv, TSC, 4, 3, atoi(argv[3])));
		if (!bits) return 4;
		FILE *o = fopen(argv[4], "wb");
		fwrite(bits->begin(), 1, bits->bytes(), o);
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	if (!strcmp(argv[1], "sch") && argc == 5) {
		/* ms_rx_lower.cpp:213-250: detectSCHBurst() then demodAnyBurst(burst, SCH, 4, &ebp) on the first 625 samples */
		std::vector<char> raw = slurp(argv[2]);
		size_t n = raw.size() / sizeof(complex);
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n, dummy_alloc, dummy_free);
		const int st = atoi(argv[3]);
		struct estim_burst_params ebp;
		memset(&ebp, 0, sizeof(ebp));
		int rc = detectSCHBurst(sv, BURST_THRESH, 4, st == 1 ? sch_detect_type::SCH_DETECT_NARROW
						     : st == 2 ? sch_detect_type::SCH_DETECT_BUFFER : sch_detect_type::SCH_DETECT_FULL, &ebp);
		FILE *o = fopen(argv[4], "w");
		fprintf(o, "rc %d\ntoa %.9g\namp %.9g %.9g\nci %.9g\n", rc, ebp.toa, ebp.amp.real(), ebp.amp.imag(), ebp.ci);
		if (rc > 0 && st == 0) {
			signalVector one(reinterpret_cast<complex *>(raw.data()), 0, 625, dummy_alloc, dummy_free);
			std::unique_ptr<SoftVector> soft(demodAnyBurst(one, SCH, 4, &ebp));
			fprintf(o, "bits ");
			for (size_t i = 0; soft && i < 148; i++) fputc(soft->bit(i) ? '1' : '0', o);
			fputc('\n', o);
		}
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	if (!strcmp(argv[1], "batch") && argc == 9) {
		std::vector<char> iq = slurp(argv[2]), pr = slurp(argv[3]);
		size_t n = atol(argv[4]);
		int sps = atoi(argv[5]);
		size_t burst_len = atol(argv[6]);
		std::vector<BurstRequest> req(n);
		const int16_t *s = reinterpret_cast<const int16_t *>(iq.data());
		for (size_t i = 0; i < n; i++) {
			const unsigned char *p = reinterpret_cast<const unsigned char *>(pr.data()) + 8 * i;
			req[i].fn = (uint32_t)i; req[i].tn = (uint8_t)(i & 7);
			req[i].iq = s + i * burst_len * 2;
			req[i].type = (CorrType)p[0];
			req[i].tsc = p[1];
			req[i].max_toa = p[2] | (p[3] << 8);
		}
		std::vector<BurstIndication> out(n);
		bool egprs = false;
		for (size_t i = 0; i < n; i++) egprs |= (req[i].type == EDGE);
		int rc = pullRadioVectorBatch(req.data(), n, sps, burst_len, 32767.0, 0.0, out.data(), egprs);
		if (rc) { fprintf(stderr, "pullRadioVectorBatch rc=%d\n", rc); return 4; }
		FILE *fr = fopen(argv[7], "wb"), *fs = fopen(argv[8], "wb");
		for (size_t i = 0; i < n; i++) {
			float rec[6] = { (float)out[i].rc, (float)out[i].toa, out[i].ci, (float)out[i].rssi, (float)out[i].idle,
					 (float)out[i].tsc };
			fwrite(rec, sizeof(rec), 1, fr);
			fwrite(out[i].rx_burst, sizeof(float), egprs ? 444 : 148, fs);
		}
		fclose(fr);
		fclose(fs);
		sigProcLibDestroy();
		return 0;
	}
	// multi <wide.s16> <n_blocks> <chans> <out_prefix>: RadioInterfaceMulti Rx in chunks of 1, 2, 3, ... blocks
	if (!strcmp(argv[1], "batchva") && argc == 7) {
		/* pullRadioVectorBatchVA: iq.s16 params.bin n out_rec.bin out_soft.bin; records {rc, toa, ci, rssi, idle, tsc, energy} */
		std::vector<char> iq = slurp(argv[2]), pr = slurp(argv[3]);
		size_t n = atol(argv[4]);
		std::vector<BurstRequest> req(n);
		for (size_t i = 0; i < n; i++) {
			const unsigned char *p = reinterpret_cast<const unsigned char *>(pr.data()) + 8 * i;
			req[i].fn = (uint32_t)i; req[i].tn = (uint8_t)(i & 7);
			req[i].iq = reinterpret_cast<const int16_t *>(iq.data()) + i * 625 * 2;
			req[i].type = (CorrType)p[0];
			req[i].tsc = p[1];
			req[i].max_toa = p[2] | (p[3] << 8);
		}
		std::vector<BurstIndication> ind(n);
		if (pullRadioVectorBatchVA(req.data(), n, 4, 625, 32767.0, 0.0, ind.data()) != 0) return 4;
		FILE *fr = fopen(argv[5], "wb"), *fs = fopen(argv[6], "wb");
		for (size_t i = 0; i < n; i++) {
			float rec[7] = { (float)ind[i].rc, (float)ind[i].toa, ind[i].ci, (float)ind[i].rssi, (float)ind[i].idle, (float)ind[i].tsc, ind[i].energy };
			fwrite(rec, sizeof(rec), 1, fr);
			fwrite(ind[i].rx_burst, sizeof(float), 148, fs);
		}
		fclose(fr); fclose(fs);
		sigProcLibDestroy();
		return 0;
	}

	// gather <iq.s16> <params.bin> <n> <chans> <max_batch> <timeout_us> <trxd_version|-1> <out.bin> [repeat [producers [fifo_depth [by_ref]]]]
	// by_ref = 1: BurstGathererConfig::by_reference -- the loaded capture is registered as the "receive ring" and the producers
	// push addresses into it (no CPU copy of the samples; the GPU fetches the gathered bursts over the link).
	// `producers` threads (default: one per channel; the reference has one RxLower thread per radio device feeding all its
	// channels) push the n bursts (burst i belongs to channel i % chans, fn = i / chans; channel c is fed by producer
	// c % producers), `chans` consumer threads pull them back (pullRadioVector's role).  out.bin: per burst, in input order,
	// {int32 code, int32 rc, float toa, float ci, float rssi, uint32 fn, uint32 tn|idle<<8|nbits<<16, uint32 pkt_len}
	// followed by 456 bytes: the TRXD datagram (trxd_version >= 0) or the first 114 soft floats (-1).
	if (!strcmp(argv[1], "gather") && argc >= 10 && argc <= 14) {
		std::vector<char> iq = slurp(argv[2]), pr = slurp(argv[3]);
		const size_t n = atol(argv[4]), chans = atol(argv[5]);
		const int repeat = argc >= 11 ? atoi(argv[10]) : 1;
		const size_t producers = argc >= 12 ? (size_t)atol(argv[11]) : chans;
		const size_t fifo_depth = argc >= 13 ? (size_t)atol(argv[12]) : 32;
		BurstGathererConfig cfg;
		memset(&cfg, 0, sizeof(cfg));
		cfg.chans = chans;
		cfg.max_batch = atol(argv[6]);
		cfg.timeout_us = atoi(argv[7]);
		cfg.fifo_depth = fifo_depth;                     /* 32 = the reference's rule (radioInterface.cpp:277) */
		cfg.sps = 4;
		cfg.burst_len = 625;
		cfg.rxFullScale = 32767.0;
		cfg.rssi_offset = 0.0;
		cfg.egprs = false;
		cfg.trxd_version = atoi(argv[8]);
		cfg.depth = 4;
		cfg.by_reference = argc >= 14 && atoi(argv[13]) != 0;
		BurstGatherer g(cfg);
		if (cfg.by_reference && !g.registerBuffer(iq.data(), iq.size())) { fprintf(stderr, "BurstGatherer::registerBuffer failed\n"); return 5; }
		if (!g.start()) { fprintf(stderr, "BurstGatherer::start failed\n"); return 5; }
		struct Rec { int32_t code, rc; float toa, ci, rssi; uint32_t fn, misc, pkt_len; uint8_t body[456]; };
		std::vector<Rec> rec(n);
		memset(rec.data(), 0, n * sizeof(Rec));
		const int16_t *s16 = reinterpret_cast<const int16_t *>(iq.data());
		std::atomic<uint64_t> retries{0};
		const auto t0 = std::chrono::steady_clock::now();
		std::vector<std::thread> th;
		for (size_t pr_i = 0; pr_i < producers; pr_i++)
			th.emplace_back([&, pr_i] {                      /* RxLower's role (radioInterface.cpp:272-291) for its channels */
				std::vector<size_t> my, cs;
				for (size_t c = pr_i; c < chans; c += producers) my.push_back(c);
				std::vector<BurstRequest> rq(my.size());
				std::vector<char> acc(my.size());
				for (int rep = 0; rep < repeat; rep++)
					for (size_t base = 0; base < n; base += chans) {     /* one "timeslot": a burst per channel */
						cs.clear();
						size_t m = 0;
						for (size_t c : my) {
							const size_t i = base + c;
							if (i >= n) break;
							const unsigned char *p = reinterpret_cast<const unsigned char *>(pr.data()) + 8 * i;
							rq[m].iq = s16 + i * 625 * 2;
							rq[m].type = (CorrType)p[0];
							rq[m].tsc = p[1];
							rq[m].max_toa = p[2] | (p[3] << 8);
							rq[m].fn = (uint32_t)(i / chans);
							rq[m].tn = (uint8_t)(i & 7);
							cs.push_back(c);
							m++;
						}
						/* a real radio would drop what does not fit; the test wants every burst back: retry the refused ones */
						size_t left = m;
						while (left) {
							g.pushSlot(cs.data(), rq.data(), left, reinterpret_cast<bool *>(acc.data()));
							size_t w = 0;
							for (size_t k = 0; k < left; k++)
								if (!acc[k]) { cs[w] = cs[k]; rq[w] = rq[k]; w++; }
							if (w) { retries += w; std::this_thread::yield(); }
							left = w;
						}
					}
			});
		for (size_t c = 0; c < chans; c++) {
			th.emplace_back([&, c] {                         /* RxUpper<c>'s role (Transceiver.cpp:1229-1253) */
				BurstIndication bi;
				uint8_t pkt[TRXD_MAX_PKT_LEN + 1];
				for (int rep = 0; rep < repeat; rep++)
					for (size_t i = c; i < n; i += chans) {
						size_t len = 0;
						Rec &r = rec[i];
						r.code = g.pull(c, &bi, cfg.trxd_version >= 0 ? pkt : NULL, &len);
						r.rc = bi.rc; r.toa = (float)bi.toa; r.ci = bi.ci; r.rssi = (float)bi.rssi; r.fn = bi.fn;
						r.misc = bi.tn | ((uint32_t)bi.idle << 8) | (bi.nbits << 16);
						r.pkt_len = (uint32_t)len;
						if (cfg.trxd_version >= 0) memcpy(r.body, pkt, len);
						else memcpy(r.body, bi.rx_burst, 456);
					}
			});
		}
		for (auto &t : th) t.join();
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		printf("gather bursts %zu seconds %.6f mbursts_per_s %.3f batches %llu push_retries %llu producers %zu consumers %zu fifo_depth %zu by_ref %d\n",
		       n * repeat, dt, n * repeat / dt * 1e-6, (unsigned long long)g.batches(), (unsigned long long)retries.load(), producers,
		       chans, fifo_depth, (int)cfg.by_reference);
		printf("devices %zu batches_per_device", g.devices());      /* TRXHIP_DEVICES=0,0: two contexts on one GPU */
		for (size_t k = 0; k < g.devices(); k++) printf(" %llu", (unsigned long long)g.batchesOn(k));
		printf("\n");
		g.stop();
		FILE *o = fopen(argv[9], "wb");
		fwrite(rec.data(), sizeof(Rec), n, o);
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	// pullrv: Transceiver::pullRadioVector(chan, bi) over the gatherer (trxPullRadioVector.h).  Burst i belongs to channel
	// i % chans (fn = i / chans, tn = i & 7); one thread pushes the slots in order, one consumer per channel pulls with its
	// own RxChanState (channel `muted` has mMuted set).  out.bin: per burst, in input order,
	// {int32 code; uint32 nbits, fn, tn, idle, modulation, tss, tsc; float ci; double rssi, toa, noise; float rx_burst[444];
	//  uint32 rx_clipping, rx_no_burst_detected}  (the channel's counters after this burst)
	if (!strcmp(argv[1], "pullrv") && argc == 9) {
		std::vector<char> iq = slurp(argv[2]), pr = slurp(argv[3]);
		const size_t n = atol(argv[4]), chans = atol(argv[5]);
		const long muted = atol(argv[6]);
		BurstGathererConfig cfg;
		memset(&cfg, 0, sizeof(cfg));
		cfg.chans = chans; cfg.max_batch = 256; cfg.timeout_us = 200; cfg.fifo_depth = 32; cfg.sps = 4; cfg.burst_len = 625;
		cfg.rxFullScale = 32767.0; cfg.rssi_offset = -3.5; cfg.egprs = false; cfg.trxd_version = -1; cfg.depth = 4;
		cfg.exact_demod = atoi(argv[7]) == 1;                       /* 0: fused, 1: bit-exact kernel, 2: cfg->use_va (Viterbi receiver) */
		cfg.use_va = atoi(argv[7]) == 2;
		BurstGatherer g(cfg);
		if (!g.start()) { fprintf(stderr, "BurstGatherer::start failed\n"); return 5; }
#pragma pack(push, 1)
		struct Rec { int32_t code; uint32_t nbits, fn, tn, idle, modulation, tss, tsc; float ci; double rssi, toa, noise; float rx[444];
			     uint32_t rx_clipping, rx_no_burst_detected; };
#pragma pack(pop)
		std::vector<Rec> rec(n);
		memset(rec.data(), 0, n * sizeof(Rec));
		const int16_t *s16 = reinterpret_cast<const int16_t *>(iq.data());
		std::vector<std::thread> th;
		th.emplace_back([&] {
			for (size_t i = 0; i < n; i++) {
				const unsigned char *p = reinterpret_cast<const unsigned char *>(pr.data()) + 8 * i;
				BurstRequest rq;
				memset(&rq, 0, sizeof(rq));
				rq.iq = s16 + i * 625 * 2; rq.type = (CorrType)p[0]; rq.tsc = p[1]; rq.max_toa = p[2] | (p[3] << 8);
				rq.fn = (uint32_t)(i / chans); rq.tn = (uint8_t)(i & 7);
				while (!g.push(i % chans, rq)) std::this_thread::yield();
			}
		});
		for (size_t c = 0; c < chans; c++)
			th.emplace_back([&, c] {
				RxChanState st;
				st.mMuted = (long)c == muted;
				struct trx_ul_burst_ind bi;
				for (size_t i = c; i < n; i += chans) {
					memset(&bi, 0xa5, sizeof(bi));                     /* every field the reference initialises must be written */
					Rec &r = rec[i];
					r.code = trxPullRadioVector(g, st, c, &bi);
					r.nbits = bi.nbits; r.fn = bi.fn; r.tn = bi.tn; r.idle = bi.idle; r.modulation = (uint32_t)bi.modulation;
					r.tss = bi.tss; r.tsc = bi.tsc; r.ci = bi.ci; r.rssi = bi.rssi; r.toa = bi.toa; r.noise = bi.noise;
					if (r.code == 0 && !bi.idle) memcpy(r.rx, bi.rx_burst, bi.nbits * sizeof(float));
					r.rx_clipping = st.ctrs.rx_clipping; r.rx_no_burst_detected = st.ctrs.rx_no_burst_detected;
				}
			});
		for (auto &t : th) t.join();
		g.stop();
		FILE *o = fopen(argv[8], "wb");
		fwrite(rec.data(), sizeof(Rec), n, o);
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	// trxdhost <ind.bin> <n> <version> <out.bin>: the host packer trxdPackBurstInd() over n indications given as
	// {float rx_burst[444]; uint32 nbits, fn, tn, idle, modulation, tss, tsc; float ci; double rssi, toa}; out: n x (u16 len + 456 bytes)
	if (!strcmp(argv[1], "trxdhost") && argc == 6) {
		std::vector<char> raw = slurp(argv[2]);
		const size_t n = atol(argv[3]);
		const unsigned ver = atoi(argv[4]);
		struct In { float rx[444]; uint32_t nbits, fn, tn, idle, modulation, tss, tsc; float ci; double rssi, toa; };
		const In *in = reinterpret_cast<const In *>(raw.data());
		FILE *o = fopen(argv[5], "wb");
		for (size_t i = 0; i < n; i++) {
			BurstIndication bi;
			memset(&bi, 0, sizeof(bi));
			memcpy(bi.rx_burst, in[i].rx, sizeof(bi.rx_burst));
			bi.nbits = in[i].nbits; bi.fn = in[i].fn; bi.tn = (uint8_t)in[i].tn; bi.idle = in[i].idle != 0;
			bi.modulation = (uint8_t)in[i].modulation; bi.tss = (uint8_t)in[i].tss; bi.tsc = (uint8_t)in[i].tsc;
			bi.ci = in[i].ci; bi.rssi = in[i].rssi; bi.toa = in[i].toa;
			uint8_t buf[460];
			memset(buf, 0, sizeof(buf));
			const int len = trxdPackBurstInd(buf, &bi, ver);
			const uint16_t l16 = (uint16_t)(len < 0 ? 0xffff : len);
			fwrite(&l16, 2, 1, o);
			fwrite(buf, 1, 456, o);
		}
		fclose(o);
		sigProcLibDestroy();
		return 0;
	}

	if (!strcmp(argv[1], "multi") && argc == 6) {
		std::vector<char> raw = slurp(argv[2]);
		size_t n_blocks = atol(argv[3]), chans = atol(argv[4]);
		MultiArfcnRx rx(chans);
		if (!rx.init()) { fprintf(stderr, "MultiArfcnRx::init failed\n"); return 5; }
		std::vector<std::vector<complex> > out;
		const int16_t *w = reinterpret_cast<const int16_t *>(raw.data());
		size_t pos = 0, step = 1;
		while (pos < n_blocks) {
			size_t nb = step < n_blocks - pos ? step : n_blocks - pos;
			if (rx.pullBuffer(w + pos * 192 * 4 * 2, nb, out)) { fprintf(stderr, "pullBuffer failed\n"); return 6; }
			pos += nb;
			step++;
		}
		for (size_t l = 0; l < chans; l++) {
			char path[512];
			snprintf(path, sizeof(path), "%s%zu.cf32", argv[5], l);
			FILE *f = fopen(path, "wb");
			fwrite(out[l].data(), sizeof(complex), out[l].size(), f);
			fclose(f);
		}
		sigProcLibDestroy();
		return 0;
	}
	return 2;
}
