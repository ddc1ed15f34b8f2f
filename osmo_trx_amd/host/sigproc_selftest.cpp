// sigproc_selftest.cpp -- exercises the sigProcLib-compatible host shim the way the reference's callers do
// (burst-gen.cpp:274-290 for the captured burst; Transceiver.cpp:768-803 for the detect -> demod -> slice
// sequence).  Reads inputs prepared by tests/test_gpu_host_shim.py, writes results as plain text/binary.
//   sigproc_selftest capture <cfile> <out.txt>
//   sigproc_selftest delay <cfile> <delay> <scale_re> <scale_im> <out.cf32>
//   sigproc_selftest batchva <iq.s16> <params.bin> <n> <rec.bin> <soft.bin>
//   sigproc_selftest va <cfile> <tsc> <out.f32>
//   sigproc_selftest sch <cfile> <0 full | 1 narrow | 2 buffer> <out.txt>
//   sigproc_selftest batch <iq.s16> <params.bin> <n> <sps> <burst_len> <out_results.bin> <out_soft.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "sigProcLib.h"
#include "MultiArfcnRx.h"

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

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	if (!sigProcLibSetup()) { fprintf(stderr, "no GPU\n"); return 3; }

	if (!strcmp(argv[1], "capture") && argc == 4) {
		std::vector<char> raw = slurp(argv[2]);
		size_t n = raw.size() / sizeof(complex);
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n);
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
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n);
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
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n);
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
		signalVector sv(reinterpret_cast<complex *>(raw.data()), 0, n);
		const int st = atoi(argv[3]);
		struct estim_burst_params ebp;
		memset(&ebp, 0, sizeof(ebp));
		int rc = detectSCHBurst(sv, BURST_THRESH, 4, st == 1 ? sch_detect_type::SCH_DETECT_NARROW
						     : st == 2 ? sch_detect_type::SCH_DETECT_BUFFER : sch_detect_type::SCH_DETECT_FULL, &ebp);
		FILE *o = fopen(argv[4], "w");
		fprintf(o, "rc %d\ntoa %.9g\namp %.9g %.9g\nci %.9g\n", rc, ebp.toa, ebp.amp.real(), ebp.amp.imag(), ebp.ci);
		if (rc > 0 && st == 0) {
			signalVector one(reinterpret_cast<complex *>(raw.data()), 0, 625);
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
