// pullrv_driver.cpp -- trxPullRadioVector() (host/trxPullRadioVector.cpp) on the CPU stand-in of the host pipeline: the host
// logic of pullRadioVector() around the DSP -- struct bi initialisation, OFF / muted / IDLE early-outs, the 20-entry noise
// ring, rssi / noise in dBFS, rate counters, return codes -- without a GPU (tests/test_pullrv_cpu.py compares every record
// with the oracle's restatement of the same function, orc_pull_radio_vector).
//   pullrv_driver <n> <chans> <muted_chan> <out.bin>
// Burst i: channel i % chans, fn = i / chans, type from a fixed pattern, "energy" (what the stub echoes as bi.energy)
// = 1 + (i * 7919) % 5003 -- the Python side regenerates the same schedule.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trxPullRadioVector.h"
#ifdef TRXHIP_SA_NS
using namespace trxhip_sa;
#endif

int main(int argc, char **argv)
{
	if (argc != 5) return 2;
	const size_t n = atol(argv[1]), chans = atol(argv[2]);
	const long muted = atol(argv[3]);
	BurstGathererConfig cfg;
	memset(&cfg, 0, sizeof(cfg));
	cfg.chans = chans; cfg.max_batch = 64; cfg.timeout_us = 100; cfg.fifo_depth = 32; cfg.sps = 4; cfg.burst_len = 625;
	cfg.rxFullScale = 32767.0; cfg.rssi_offset = -3.5; cfg.trxd_version = -1; cfg.depth = 4;
	BurstGatherer g(cfg);
	if (!g.start()) return 3;
#pragma pack(push, 1)
	struct Rec { int32_t code; uint32_t nbits, fn, tn, idle, modulation, tss, tsc; float ci; double rssi, toa, noise; float rx0;
		     uint32_t rx_clipping, rx_no_burst_detected; };
#pragma pack(pop)
	std::vector<Rec> rec(n);
	memset(rec.data(), 0, n * sizeof(Rec));
	std::vector<std::thread> th;
	th.emplace_back([&] {
		std::vector<int16_t> iq(625 * 2, 0);
		for (size_t i = 0; i < n; i++) {
			const uint32_t fn = (uint32_t)(i / chans);
			iq[0] = (int16_t)(fn & 0x7fff); iq[1] = (int16_t)(fn >> 15);
			iq[2] = (int16_t)(1 + (i * 7919) % 5003);                 /* echoed as bi.energy by the stub */
			BurstRequest rq;
			memset(&rq, 0, sizeof(rq));
			rq.iq = iq.data();
			rq.type = (i % 11 == 3) ? OFF : (i % 5 == 1) ? IDLE : (i % 7 == 2) ? RACH : TSC;
			rq.tsc = (unsigned)(i & 7); rq.max_toa = 3; rq.fn = fn; rq.tn = (uint8_t)(i & 7);
			while (!g.push(i % chans, rq)) std::this_thread::yield();
		}
	});
	for (size_t c = 0; c < chans; c++)
		th.emplace_back([&, c] {
			RxChanState st;
			st.mMuted = (long)c == muted;
			struct trx_ul_burst_ind bi;
			for (size_t i = c; i < n; i += chans) {
				memset(&bi, 0xa5, sizeof(bi));
				Rec &r = rec[i];
				r.code = trxPullRadioVector(g, st, c, &bi);
				r.nbits = bi.nbits; r.fn = bi.fn; r.tn = bi.tn; r.idle = bi.idle; r.modulation = (uint32_t)bi.modulation;
				r.tss = bi.tss; r.tsc = bi.tsc; r.ci = bi.ci; r.rssi = bi.rssi; r.toa = bi.toa; r.noise = bi.noise;
				r.rx0 = (r.code == 0 && !bi.idle) ? bi.rx_burst[0] : -1.0f;
				r.rx_clipping = st.ctrs.rx_clipping; r.rx_no_burst_detected = st.ctrs.rx_no_burst_detected;
			}
		});
	for (auto &t : th) t.join();
	g.stop();
	FILE *o = fopen(argv[4], "wb");
	fwrite(rec.data(), sizeof(Rec), n, o);
	fclose(o);
	return 0;
}
