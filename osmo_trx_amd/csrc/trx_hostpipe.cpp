// trx_hostpipe.cpp -- host-fed, stream-pipelined form of the hot path (include/trxhip.h, trxhip_hostpipe_*).
//
// pullRadioVector()'s callers (Transceiver.cpp:665-815) hold bursts in host memory; this is the piece between
// them and trxhip_detect_demod_batch(): `depth` staging slots, each = pinned host buffers + device buffers + one
// HIP stream.  A submitted slot runs H2D -> detect/demod [-> TRXD wire packer] -> D2H on its own stream, so the
// upload of one slot, the kernels of another and the download of a third overlap (the two DMA directions and the
// compute queue are independent engines).  Everything is allocated in create(); submit() only enqueues.
//
// Round 3: a batch is FIVE enqueues, not ten.  A slot's inputs are one pinned block and one device block with the same
// layout ([params][meta][bursts]: one H2D copy of the used prefix), its results and soft rows one device block and one
// pinned block ([results][soft rows]: one D2H copy), and the TRXD packer writes datagrams and lengths straight into the
// pinned buffers (they are device-visible; 160 bytes per burst over the link, posted writes).  A round trip of a
// 256-burst batch through an idle pipe went from 86-100 us to what tools/bench_hostpipe_rt.py prints now -- the gather
// stage under the reference's 32-deep FIFO rule is bound by exactly this latency.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "trx_ctx.h"

struct trxhip_hostpipe {
	trxhip_ctx *ctx;
	trxhip_hostpipe_cfg cfg;
	int dev_soft_stride;               /* stride of the device-side soft rows (>= what the packer needs) */
	size_t in_iq_off, in_meta_off;     /* [params][meta][bursts] inside the input blocks */
	size_t out_soft_off;               /* [results][soft rows] inside the output blocks */
	struct Region {                    /* host ranges registered for bursts by reference */
		const char *base;
		size_t bytes;
		const char *dev_base;          /* the range's address on this pipe's device */
		bool owned;                    /* pinned by this pipe (hipHostRegister): released by it */
	} region[8];
	int n_region;
	struct Slot {
		hipStream_t stream;
		hipEvent_t done;
		bool busy, failed;
		size_t n;
		trxhip_hostpipe_slot h;        /* pinned host (views into h_in / h_out; pkt, pkt_len separate) */
		char *h_in, *d_in;             /* one block each: params, meta, bursts */
		char *dv_in;                   /* device-side address of the pinned input block (small batches are read in place) */
		char *h_out, *d_out;           /* one block each: results, soft rows */
		int16_t *d_iq;
		int16_t *d_iq_sel;             /* n_paths > 1: the chosen path of every burst */
		float *d_avg;                  /* n_paths > 1: path-averaged energy; use_va: energy of the burst as read */
		int16_t *d_shift;              /* use_va: the bursts shifted by 20 samples (what the detector looks at) */
		float *d_cf;                   /* use_va: the bursts as complex64 (what the Viterbi receiver reads) */
		trxhip_burst_params *d_params;
		trxhip_trxd_meta *d_meta;
		trxhip_burst_result *d_results;
		float *d_soft;
		uint8_t *dv_pkt;               /* device-side addresses of the pinned h.pkt / h.pkt_len */
		uint16_t *dv_pkt_len;
		trxhip_burst_result *dv_results; /* ... of h.results (TRXD-only pipes: the packer delivers the records too) */
		const int16_t **h_src;         /* by reference: the caller's host pointers (pinned, max_bursts) ... */
		unsigned long long *h_src_dev, *dv_src_dev;   /* ... and their device-side addresses, read in place by the gather kernel */
	} slot[16];
};

/* pinned, device-mapped and coherent, stated explicitly: the kernels read small batches in place and the TRXD packer writes
 * datagrams straight into these buffers -- that must not hang on a runtime default (HIP_HOST_COHERENT) */
static bool pin(void **p, size_t bytes)
{
	return hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
}
static bool dev(void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16) == hipSuccess; }

extern "C" {

int trxhip_hostpipe_create(trxhip_ctx *ctx, const trxhip_hostpipe_cfg *c, trxhip_hostpipe **out)
{
	if (!ctx || !c || !out)
		return TRXHIP_EINVAL;
	if (c->max_bursts < 1 || c->max_bursts > (1u << 24) || c->depth < 2 || c->depth > 16)
		return TRXHIP_EINVAL;
	if ((c->sps != 1 && c->sps != 4) || c->burst_len < 148 || c->burst_len > TRXHIP_MAX_BURST_LEN)
		return TRXHIP_EINVAL;
	if (c->soft_stride < 0 || (c->soft_stride > 0 && c->soft_stride < 148))
		return TRXHIP_EINVAL;
	if (c->pkt_stride < 0 || (c->pkt_stride > 0 && (c->pkt_stride < 160 || (c->pkt_stride & 3))))
		return TRXHIP_EINVAL;
	if (c->soft_stride == 0 && c->pkt_stride == 0)
		return TRXHIP_EINVAL;
	if (c->n_paths < 0 || c->n_paths > 8)
		return TRXHIP_EINVAL;
	const bool use_va = (c->flags & TRXHIP_FLAG_USE_VA) != 0;
	if (use_va && (c->sps != 4 || c->burst_len <= 40 || (c->flags & TRXHIP_FLAG_EXACT_DEMOD)))
		return TRXHIP_EINVAL;                                  /* grgsm_vitac is a 4-SPS receiver */
	if (c->n_paths > 1) {                                      /* the diversity energy scan must stay inside a path (trx_capi.cpp) */
		const int window = 20 * c->sps < c->burst_len ? 20 * c->sps : c->burst_len;
		if (4 * (window - 1) >= c->burst_len)
			return TRXHIP_EINVAL;
	}
	if (with_device(ctx))
		return TRXHIP_EIO;

	trxhip_hostpipe *p = new (std::nothrow) trxhip_hostpipe();
	if (!p)
		return TRXHIP_ENOMEM;
	memset(p, 0, sizeof(*p));
	p->ctx = ctx;
	p->cfg = *c;
	if (c->pkt_stride)
		p->cfg.flags |= TRXHIP_FLAG_SLICE;                     /* the packer quantises vectorSlicer()'s 0..1 values */
	/* device rows: what the caller downloads, or -- TRXD only -- what the packer can consume */
	p->dev_soft_stride = c->soft_stride ? c->soft_stride : (c->pkt_stride >= TRXHIP_TRXD_V1_HDR + 444 ? 444 : 148);
	const size_t nb = c->max_bursts;
	const size_t np = c->n_paths > 1 ? (size_t)c->n_paths : 1;
	auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
	p->in_meta_off = up(nb * sizeof(trxhip_burst_params));
	p->in_iq_off = p->in_meta_off + (c->pkt_stride ? up(nb * sizeof(trxhip_trxd_meta)) : 0);
	const size_t in_bytes = p->in_iq_off + nb * np * c->burst_len * 4;
	p->out_soft_off = up(nb * sizeof(trxhip_burst_result));
	const size_t out_bytes_h = p->out_soft_off + (c->soft_stride ? nb * c->soft_stride * sizeof(float) : 0);
	const size_t out_bytes_d = p->out_soft_off + nb * p->dev_soft_stride * sizeof(float);
	bool ok = true;
	for (int s = 0; s < c->depth && ok; s++) {
		trxhip_hostpipe::Slot &sl = p->slot[s];
		ok = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) == hipSuccess &&
		     hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess &&
		     pin((void **)&sl.h_in, in_bytes) && dev((void **)&sl.d_in, in_bytes) &&
		     hipHostGetDevicePointer((void **)&sl.dv_in, sl.h_in, 0) == hipSuccess &&
		     pin((void **)&sl.h_out, out_bytes_h) && dev((void **)&sl.d_out, out_bytes_d) &&
		     pin((void **)&sl.h_src, nb * sizeof(void *)) && pin((void **)&sl.h_src_dev, nb * sizeof(unsigned long long)) &&
		     hipHostGetDevicePointer((void **)&sl.dv_src_dev, sl.h_src_dev, 0) == hipSuccess &&
		     (np == 1 || (dev((void **)&sl.d_iq_sel, nb * c->burst_len * 4) && dev((void **)&sl.d_avg, nb * sizeof(float)))) &&
		     (!use_va || (dev((void **)&sl.d_shift, nb * c->burst_len * 4) && dev((void **)&sl.d_cf, nb * c->burst_len * 8) &&
				  (sl.d_avg || dev((void **)&sl.d_avg, nb * sizeof(float)))));
		if (!ok)
			break;
		sl.h.params = reinterpret_cast<trxhip_burst_params *>(sl.h_in);
		sl.h.meta = c->pkt_stride ? reinterpret_cast<trxhip_trxd_meta *>(sl.h_in + p->in_meta_off) : nullptr;
		sl.h.iq = reinterpret_cast<int16_t *>(sl.h_in + p->in_iq_off);
		sl.d_params = reinterpret_cast<trxhip_burst_params *>(sl.d_in);
		sl.d_meta = c->pkt_stride ? reinterpret_cast<trxhip_trxd_meta *>(sl.d_in + p->in_meta_off) : nullptr;
		sl.d_iq = reinterpret_cast<int16_t *>(sl.d_in + p->in_iq_off);
		sl.h.results = reinterpret_cast<trxhip_burst_result *>(sl.h_out);
		sl.h.soft = c->soft_stride ? reinterpret_cast<float *>(sl.h_out + p->out_soft_off) : nullptr;
		sl.d_results = reinterpret_cast<trxhip_burst_result *>(sl.d_out);
		sl.d_soft = reinterpret_cast<float *>(sl.d_out + p->out_soft_off);
		if (c->pkt_stride)
			ok = pin((void **)&sl.h.pkt, nb * c->pkt_stride) && pin((void **)&sl.h.pkt_len, nb * sizeof(uint16_t)) &&
			     hipHostGetDevicePointer((void **)&sl.dv_pkt, sl.h.pkt, 0) == hipSuccess &&
			     hipHostGetDevicePointer((void **)&sl.dv_pkt_len, sl.h.pkt_len, 0) == hipSuccess &&
			     hipHostGetDevicePointer((void **)&sl.dv_results, sl.h.results, 0) == hipSuccess;
	}
	if (!ok) {
		trxhip_hostpipe_destroy(p);
		return TRXHIP_ENOMEM;
	}
	*out = p;
	return TRXHIP_OK;
}

void trxhip_hostpipe_destroy(trxhip_hostpipe *p)
{
	if (!p)
		return;
	(void)with_device(p->ctx);
	for (int s = 0; s < 16; s++) {
		trxhip_hostpipe::Slot &sl = p->slot[s];
		if (sl.stream) (void)hipStreamSynchronize(sl.stream);
		void *hp[] = { sl.h_in, sl.h_out, sl.h.pkt, sl.h.pkt_len, (void *)sl.h_src, sl.h_src_dev };
		for (void *q : hp) if (q) (void)hipHostFree(q);
		void *dp[] = { sl.d_in, sl.d_out, sl.d_iq_sel, sl.d_avg, sl.d_shift, sl.d_cf };
		for (void *q : dp) if (q) (void)hipFree(q);
		if (sl.done) (void)hipEventDestroy(sl.done);
		if (sl.stream) (void)hipStreamDestroy(sl.stream);
	}
	for (int r = 0; r < p->n_region; r++)
		if (p->region[r].owned)
			(void)hipHostUnregister(const_cast<char *>(p->region[r].base));
	delete p;
}

int trxhip_hostpipe_slot_buffers(trxhip_hostpipe *p, int slot, trxhip_hostpipe_slot *out)
{
	if (!p || !out || slot < 0 || slot >= p->cfg.depth)
		return TRXHIP_EINVAL;
	*out = p->slot[slot].h;
	return TRXHIP_OK;
}

int trxhip_hostpipe_set_levels(trxhip_hostpipe *p, float threshold, float full_scale, float rssi_offset)
{
	if (!p || !(full_scale > 0.0f) || !(threshold >= 0.0f))
		return TRXHIP_EINVAL;
	p->cfg.threshold = threshold;
	p->cfg.full_scale = full_scale;
	p->cfg.rssi_offset = rssi_offset;
	return TRXHIP_OK;
}

#define TRX_RUN_MIN 16                   /* bursts (40 KB at 4 SPS): below this the copy engine's start-up cost exceeds the kernel's fetch */
#define TRX_RUN_MAX 64                   /* copies per batch */
static int submit_slot(trxhip_hostpipe *p, int slot, size_t n, bool by_ref)
{
	if (!p || slot < 0 || slot >= p->cfg.depth || n > p->cfg.max_bursts)
		return TRXHIP_EINVAL;
	trxhip_hostpipe::Slot &sl = p->slot[slot];
	if (sl.busy)
		return TRXHIP_EINVAL;                                  /* wait() first */
	const trxhip_hostpipe_cfg &c = p->cfg;
	const size_t np = c.n_paths > 1 ? (size_t)c.n_paths : 1;
	struct Run { size_t first, len, step; } runs[TRX_RUN_MAX];
	int n_runs = 0;
	size_t n_left = 0;
	if (by_ref) {
		/* host pointer -> device-side address, range by range (the last hit first: a radio has one ring); nothing is enqueued
		 * unless every burst lies inside a registered range */
		const size_t burst_bytes = np * (size_t)c.burst_len * 4;
		int r = 0;
		for (size_t i = 0; i < n; i++) {
			const char *q = reinterpret_cast<const char *>(sl.h_src[i]);
			bool in = false;
			for (int k = 0; k < p->n_region && !in; k++) {
				const trxhip_hostpipe::Region &g = p->region[(r + k) % p->n_region];
				if (q >= g.base && (size_t)(q - g.base) <= g.bytes && g.bytes - (size_t)(q - g.base) >= burst_bytes) {
					r = (r + k) % p->n_region;
					in = true;
				}
			}
			if (!in || (reinterpret_cast<uintptr_t>(q) & 3u))
				return TRXHIP_EINVAL;
			sl.h_src_dev[i] = (unsigned long long)reinterpret_cast<uintptr_t>(p->region[r].dev_base + (q - p->region[r].base));
		}
		/* Run coalescing: a radio cuts the consecutive bursts of a channel from one contiguous buffer (radioInterface.cpp:272-291),
		 * so the pointer list of a batch is a handful of runs -- addresses a constant step apart -- not n scattered ones.  A run
		 * of TRX_RUN_MIN bursts or more goes to the copy engine (one hipMemcpyAsync, or one hipMemcpy2DAsync when the step is
		 * wider than a burst: 55 GB/s against the 45 a kernel reads the link at, a cache line per request); the fetch kernel
		 * keeps what is left and skips the bursts whose address is zeroed here.  At most TRX_RUN_MAX copies per batch (each
		 * costs the host a few microseconds): further runs stay with the kernel. */
		n_runs = 0;
		n_left = n;
		for (size_t i = 0; i + 1 < n && n_runs < TRX_RUN_MAX;) {
			const uintptr_t a0 = reinterpret_cast<uintptr_t>(sl.h_src[i]), a1 = reinterpret_cast<uintptr_t>(sl.h_src[i + 1]);
			const size_t step = a1 > a0 ? a1 - a0 : 0;
			size_t j = i + 1;
			if (step >= burst_bytes && step <= (1u << 20))          /* (forward, no overlap; the 2-D copy's pitch is bounded) */
				while (j + 1 < n && reinterpret_cast<uintptr_t>(sl.h_src[j + 1]) - reinterpret_cast<uintptr_t>(sl.h_src[j]) == step &&
				       reinterpret_cast<uintptr_t>(sl.h_src[j + 1]) > reinterpret_cast<uintptr_t>(sl.h_src[j]))
					j++;
			else
				j = i;
			const size_t len = j - i + 1;
			if (len >= TRX_RUN_MIN) {
				/* (every burst of the run was found inside a registered range above; a run that spans two ranges is still two
				 * valid host ranges to the copy engine, which takes the HOST addresses) */
				runs[n_runs++] = { i, len, step };
				for (size_t k = i; k <= j; k++)
					sl.h_src_dev[k] = 0ull;
				n_left -= len;
				i = j + 1;
			} else {
				i = j > i ? j : i + 1;
			}
		}
	}
	sl.n = n;
	sl.failed = false;
	if (n == 0)
		return TRXHIP_OK;
	if (with_device(p->ctx))
		return TRXHIP_EIO;
	hipStream_t st = sl.stream;
	/* one upload: [params][meta][the n bursts] -- or none: a small batch is read by the kernels where it lies (pinned memory
	 * is device-visible; the detector fetches every sample once, one burst ahead, so the link's latency is covered the way
	 * HBM's is and the copy engine's start-up cost -- more than such a transfer itself -- is not paid) */
	static const size_t zc_max = getenv("TRXHIP_HOSTPIPE_ZC") ? (size_t)atol(getenv("TRXHIP_HOSTPIPE_ZC")) : 1024;   /* bursts x paths */
	/* (by reference: the parameters and the TRXD meta -- 16 bytes per burst -- are read where they lie as well: a second
	 * copy-engine transfer per batch in front of the run(s) cost 3 % of the link, profiles/r06_gather.txt) */
	const bool in_place = by_ref || n * np <= zc_max;
	const char *const in = in_place ? sl.dv_in : sl.d_in;
	bool ok = in_place ||
		  hipMemcpyAsync(sl.d_in, sl.h_in, p->in_iq_off + n * np * c.burst_len * 4, hipMemcpyHostToDevice, st) == hipSuccess;
	int rc = ok ? TRXHIP_OK : TRXHIP_EIO;
	if (rc == TRXHIP_OK && by_ref) {
		const size_t burst_bytes = np * (size_t)c.burst_len * 4;
		for (int k = 0; k < n_runs && rc == TRXHIP_OK; k++) {      /* the runs: the copy engine, from the host addresses */
			char *const dst = reinterpret_cast<char *>(sl.d_iq) + runs[k].first * burst_bytes;
			const void *const src = reinterpret_cast<const void *>(sl.h_src[runs[k].first]);
			const hipError_t e = runs[k].step == burst_bytes
				? hipMemcpyAsync(dst, src, runs[k].len * burst_bytes, hipMemcpyHostToDevice, st)
				: hipMemcpy2DAsync(dst, burst_bytes, src, runs[k].step, burst_bytes, runs[k].len, hipMemcpyHostToDevice, st);
			if (e != hipSuccess)
				rc = TRXHIP_EIO;
		}
		if (rc == TRXHIP_OK && n_left)                             /* the rest, fetched through their pointers by the device */
			rc = trx_launch_gather_bursts(sl.dv_src_dev, sl.d_iq, n, (unsigned)(np * (size_t)c.burst_len), st);
	}
	const int16_t *d_bursts = by_ref ? sl.d_iq : reinterpret_cast<const int16_t *>(in + p->in_iq_off);
	const trxhip_burst_params *const d_params = reinterpret_cast<const trxhip_burst_params *>(in);
	const trxhip_trxd_meta *const d_meta = reinterpret_cast<const trxhip_trxd_meta *>(in + p->in_meta_off);
	if (rc == TRXHIP_OK && np > 1) {                              /* Transceiver.cpp:723-741: the path with the highest energy */
		rc = trxhip_select_diversity_batch(p->ctx, d_bursts, n, (int)np, c.burst_len, c.sps, sl.d_iq_sel, sl.d_avg, nullptr, st);
		d_bursts = sl.d_iq_sel;
	}
	if (c.flags & TRXHIP_FLAG_USE_VA) {
		/* cfg->use_va (Transceiver.cpp:760-787): detection on the copy shifted by 20 samples -- samples 20 .. len - 20 in
		 * front, zeros behind (shift_vec, :679, :762) -- no soft bits from it; power of the burst as read; the Viterbi
		 * receiver on the unshifted, scaled burst for the slots the detector found */
		const size_t bb = (size_t)c.burst_len * 4, keep = ((size_t)c.burst_len - 40) * 4;
		if (rc == TRXHIP_OK &&
		    (hipMemsetAsync(sl.d_shift, 0, n * bb, st) != hipSuccess ||
		     hipMemcpy2DAsync(sl.d_shift, bb, d_bursts + 40, bb, keep, n, hipMemcpyDeviceToDevice, st) != hipSuccess))
			rc = TRXHIP_EIO;
		if (rc == TRXHIP_OK)
			rc = trxhip_detect_demod_batch(p->ctx, sl.d_shift, d_params, sl.d_results, nullptr, n, c.burst_len, c.sps,
						       c.threshold, c.full_scale, p->dev_soft_stride,
						       c.flags & ~(TRXHIP_FLAG_USE_VA | TRXHIP_FLAG_SLICE), st);
		if (rc == TRXHIP_OK)
			rc = trxhip_convert_short_float(p->ctx, sl.d_cf, d_bursts, n * (size_t)c.burst_len * 2, st);
		if (rc == TRXHIP_OK && np == 1)                           /* (with diversity the path average is there already) */
			rc = trxhip_energy_detect_batch_cf32(p->ctx, sl.d_cf, n, c.burst_len, 20u * (unsigned)c.sps, sl.d_avg, st);
		if (rc == TRXHIP_OK)
			rc = trxhip_demod_va_batch_cf32(p->ctx, sl.d_cf, d_params, sl.d_results, sl.d_soft, nullptr, n, c.burst_len,
							1.0f / (float)((1 << 14) - 1), p->dev_soft_stride, c.flags & TRXHIP_FLAG_SLICE, st);
		if (rc == TRXHIP_OK)                                      /* energy / rssi of the record: the burst as read (:724-751) */
			rc = trxhip_apply_diversity_power(p->ctx, sl.d_results, d_params, sl.d_avg, n, c.full_scale, st);
	} else {
		/* the slot types are known here (the host's copy of the parameters): a batch with few normal-burst slots goes to the
		 * general kernel alone (include/trxhip.h, TRXHIP_FLAG_FEW_NB_SLOTS) */
		size_t n_nb = 0;
		for (size_t i = 0; i < n; i++)
			n_nb += sl.h.params[i].type == TRXHIP_TSC && sl.h.params[i].tsc < 8 && sl.h.params[i].max_toa <= 32;
		if (rc == TRXHIP_OK)
			rc = trxhip_detect_demod_batch(p->ctx, d_bursts, d_params, sl.d_results, sl.d_soft, n, c.burst_len, c.sps,
						       c.threshold, c.full_scale, p->dev_soft_stride,
						       c.flags | (32 * (n - n_nb) > n ? TRXHIP_FLAG_FEW_NB_SLOTS : 0), st);
		if (rc == TRXHIP_OK && np > 1)                            /* :741, :751: rssi from the path average */
			rc = trxhip_apply_diversity_power(p->ctx, sl.d_results, d_params, sl.d_avg, n, c.full_scale, st);
	}
	const bool records_by_packer = c.pkt_stride && !c.soft_stride;
	if (rc == TRXHIP_OK && c.pkt_stride)                          /* datagrams and lengths: straight into the pinned buffers */
		rc = trx_launch_pack_trxd_wire(sl.d_results, d_params, sl.d_soft, p->dev_soft_stride, d_meta, sl.dv_pkt, c.pkt_stride,
					       sl.dv_pkt_len, n, c.rssi_offset, st, records_by_packer ? sl.dv_results : nullptr);   /* (strides checked in create()) */
	/* one download: [results][the n soft rows] (none when the packer has delivered the records) */
	ok = rc == TRXHIP_OK &&
	     (records_by_packer ||
	      hipMemcpyAsync(sl.h_out, sl.d_out, c.soft_stride ? p->out_soft_off + n * c.soft_stride * sizeof(float)
							        : n * sizeof(trxhip_burst_result), hipMemcpyDeviceToHost, st) == hipSuccess);
	if (ok)
		ok = hipEventRecord(sl.done, st) == hipSuccess;
	sl.busy = true;                                            /* even on failure: wait() drains what was enqueued */
	sl.failed = !ok;
	return ok ? TRXHIP_OK : (rc != TRXHIP_OK ? rc : TRXHIP_EIO);
}

int trxhip_hostpipe_submit(trxhip_hostpipe *p, int slot, size_t n) { return submit_slot(p, slot, n, false); }
int trxhip_hostpipe_submit_by_ref(trxhip_hostpipe *p, int slot, size_t n) { return submit_slot(p, slot, n, true); }

int trxhip_hostpipe_slot_sources(trxhip_hostpipe *p, int slot, const int16_t ***out)
{
	if (!p || !out || slot < 0 || slot >= p->cfg.depth)
		return TRXHIP_EINVAL;
	*out = p->slot[slot].h_src;
	return TRXHIP_OK;
}

int trxhip_hostpipe_register_host(trxhip_hostpipe *p, const void *base, size_t bytes)
{
	if (!p || !base || bytes == 0)
		return TRXHIP_EINVAL;
	if (p->n_region >= 8)
		return TRXHIP_ENOMEM;
	for (int r = 0; r < p->n_region; r++)
		if (p->region[r].base == base)
			return TRXHIP_EINVAL;
	if (with_device(p->ctx))
		return TRXHIP_EIO;
	/* portable + mapped: every device of the process may read it.  Already registered (another pipe of a multi-device
	 * gatherer, or the caller itself): shared, not owned */
	const hipError_t e = hipHostRegister(const_cast<void *>(base), bytes, hipHostRegisterPortable | hipHostRegisterMapped);
	if (e != hipSuccess)
		(void)hipGetLastError();
	if (e != hipSuccess && e != hipErrorHostMemoryAlreadyRegistered)
		return TRXHIP_EIO;
	void *dv = nullptr;
	if (hipHostGetDevicePointer(&dv, const_cast<void *>(base), 0) != hipSuccess) {
		(void)hipGetLastError();
		if (e == hipSuccess)
			(void)hipHostUnregister(const_cast<void *>(base));
		return TRXHIP_EIO;
	}
	if (e == hipErrorHostMemoryAlreadyRegistered) {
		/* someone else's registration: it must be mapped into the device and cover the WHOLE range -- the fetch kernel and the
		 * copy engine would fault on the first burst behind its end otherwise.  Checked on the flags and on the
		 * device address of the range's last byte, which must continue the first one's. */
		unsigned fl = 0;
		void *dv_last = nullptr;
		const char *const last = static_cast<const char *>(base) + (bytes - 1);
		if (hipHostGetFlags(&fl, const_cast<void *>(base)) != hipSuccess || !(fl & hipHostRegisterMapped) ||
		    hipHostGetDevicePointer(&dv_last, const_cast<char *>(last), 0) != hipSuccess ||
		    static_cast<char *>(dv_last) != static_cast<char *>(dv) + (bytes - 1)) {
			(void)hipGetLastError();
			return TRXHIP_EINVAL;
		}
	}
	trxhip_hostpipe::Region &g = p->region[p->n_region++];
	g.base = static_cast<const char *>(base);
	g.bytes = bytes;
	g.dev_base = static_cast<const char *>(dv);
	g.owned = e == hipSuccess;
	return TRXHIP_OK;
}

int trxhip_hostpipe_unregister_host(trxhip_hostpipe *p, const void *base)
{
	if (!p || !base)
		return TRXHIP_EINVAL;
	for (int r = 0; r < p->n_region; r++)
		if (p->region[r].base == base) {
			for (int s = 0; s < p->cfg.depth; s++)
				if (p->slot[s].busy)
					return TRXHIP_EINVAL;                      /* wait() first: a slot in flight may refer to the range */
			(void)with_device(p->ctx);
			if (p->region[r].owned)
				(void)hipHostUnregister(const_cast<void *>(base));
			p->region[r] = p->region[--p->n_region];
			return TRXHIP_OK;
		}
	return TRXHIP_EINVAL;
}

int trxhip_hostpipe_wait(trxhip_hostpipe *p, int slot)
{
	if (!p || slot < 0 || slot >= p->cfg.depth)
		return TRXHIP_EINVAL;
	trxhip_hostpipe::Slot &sl = p->slot[slot];
	if (!sl.busy)
		return TRXHIP_OK;
	(void)with_device(p->ctx);                                 /* (a multi-device gatherer waits for pipes of several GPUs from one thread) */
	const bool ok = hipStreamSynchronize(sl.stream) == hipSuccess && !sl.failed;
	sl.busy = false;
	return ok ? TRXHIP_OK : TRXHIP_EIO;
}

int trxhip_hostpipe_query(trxhip_hostpipe *p, int slot)
{
	if (!p || slot < 0 || slot >= p->cfg.depth)
		return TRXHIP_EINVAL;
	trxhip_hostpipe::Slot &sl = p->slot[slot];
	if (!sl.busy || sl.failed)
		return 0;
	(void)with_device(p->ctx);
	const hipError_t e = hipEventQuery(sl.done);
	return e == hipSuccess ? 0 : (e == hipErrorNotReady ? 1 : TRXHIP_EIO);
}

int trxhip_hostpipe_run(trxhip_hostpipe *p, const int16_t *h_iq, const trxhip_burst_params *h_params,
			const trxhip_trxd_meta *h_meta, trxhip_burst_result *h_results, float *h_soft, uint8_t *h_pkt,
			uint16_t *h_pkt_len, size_t n)
{
	if (!p || (n && (!h_iq || !h_params || !h_results)))
		return TRXHIP_EINVAL;
	const trxhip_hostpipe_cfg &c = p->cfg;
	if ((h_soft && !c.soft_stride) || ((h_pkt || h_pkt_len) && !c.pkt_stride) || (c.pkt_stride && n && !h_meta))
		return TRXHIP_EINVAL;
	for (int s = 0; s < c.depth; s++)
		if (p->slot[s].busy)
			return TRXHIP_EINVAL;
	const size_t chunk = c.max_bursts, burst_bytes = (size_t)c.burst_len * 4 * (c.n_paths > 1 ? (size_t)c.n_paths : 1);
	const size_t n_chunks = (n + chunk - 1) / chunk;
	int rc = TRXHIP_OK;
	/* chunk k uses slot k % depth: stage k, submit k, then collect chunk k - depth + 1 (oldest in flight) */
	auto collect = [&](size_t k) {
		const int s = (int)(k % c.depth);
		const size_t off = k * chunk, m = (off + chunk <= n) ? chunk : n - off;
		const int r = trxhip_hostpipe_wait(p, s);
		if (r != TRXHIP_OK) { rc = r; return; }
		const trxhip_hostpipe_slot &h = p->slot[s].h;
		memcpy(h_results + off, h.results, m * sizeof(trxhip_burst_result));
		if (h_soft) memcpy(h_soft + off * c.soft_stride, h.soft, m * c.soft_stride * sizeof(float));
		if (h_pkt) memcpy(h_pkt + off * c.pkt_stride, h.pkt, m * (size_t)c.pkt_stride);
		if (h_pkt_len) memcpy(h_pkt_len + off, h.pkt_len, m * sizeof(uint16_t));
	};
	for (size_t k = 0; k < n_chunks; k++) {
		if (k >= (size_t)c.depth)
			collect(k - c.depth);
		const int s = (int)(k % c.depth);
		const size_t off = k * chunk, m = (off + chunk <= n) ? chunk : n - off;
		const trxhip_hostpipe_slot &h = p->slot[s].h;
		memcpy(h.iq, reinterpret_cast<const char *>(h_iq) + off * burst_bytes, m * burst_bytes);
		memcpy(h.params, h_params + off, m * sizeof(trxhip_burst_params));
		if (c.pkt_stride) memcpy(h.meta, h_meta + off, m * sizeof(trxhip_trxd_meta));
		const int r = trxhip_hostpipe_submit(p, s, m);
		if (r != TRXHIP_OK && rc == TRXHIP_OK) rc = r;
	}
	for (size_t k = (n_chunks > (size_t)c.depth ? n_chunks - c.depth : 0); k < n_chunks; k++)
		collect(k);
	for (int s = 0; s < c.depth; s++)                              /* after an error: leave no slot busy */
		(void)trxhip_hostpipe_wait(p, s);
	return rc;
}

}  // extern "C"
