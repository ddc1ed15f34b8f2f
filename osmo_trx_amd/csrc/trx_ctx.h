// trx_ctx.h -- the context object behind trxhip_ctx* (internal to csrc/; the C ABI only hands out the pointer).
#ifndef TRX_CTX_H
#define TRX_CTX_H
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include "../../include/trxhip.h"
#include "trx_tables.h"

struct trxhip_ctx {
	int device;
	int n_cu;
	trx_tables *d_tables;
	int no_unit;        /* tables do not have the compiled-in unit structure: keep the multiplying correlation */
	int sch_unit;       /* the SCH sequence has the unit structure the kernel's compiled-in mask expects (trx_sch.hip) */
	int no_sym;         /* decimator taps not bitwise symmetric: the kernels' straight-line paths (mirrored taps) are off */
	int no_fast;        /* sinc LUT row sums above TRX_FAST_W: the fused kernels' FAST detector (proven margins) is off */
	/* cross-die work pool of the 4-SPS kernel (trx_kernel4.hip): TRX_POOL_SLOTS counter pairs {drawn, workgroups done} of 64
	 * bytes each, handed out round-robin PER LAUNCH.  The kernel re-arms its pair itself -- the last workgroup to finish
	 * zeroes both words -- so there is no memset in front of a launch: nothing that two host threads launching on one stream
	 * could interleave (round 4's per-stream counter was zeroed by a hipMemsetAsync queued separately from the launch: memset A,
	 * memset B, kernel A, kernel B left kernel B without a pool), no table of streams that fills up, and a launch
	 * captured into a HIP graph is complete in itself (replays of one graph serialise).  Two launches share a pair only when
	 * they are TRX_POOL_SLOTS pooled launches apart and still overlap -- 1024 outstanding batches of >= 32768 bursts. */
	unsigned *d_pool;
	int pool_enabled;                  /* trxhip_set_work_pool(); default 1, 0 when TRXHIP_NO_POOL is set at creation */
	std::atomic<unsigned> pool_next;
	/* The normal-burst kernel (trx_kernel_nb.hip) and the list of bursts it leaves to the general kernel: TRX_REDO_SLOTS device
	 * buffers of TRX_REDO_HDR words + one flag byte per burst, handed out round-robin per launch under redo_mu.  A slot's header is zero
	 * between launches (the general kernel's last workgroup re-arms it); its event marks the end of the launch pair that used
	 * it last -- a slot is only handed out again (or re-allocated larger) once that event has completed. */
	int nb_enabled;                    /* trxhip_set_nb_kernel(); default 1, 0 when TRXHIP_NO_NB_KERNEL is set at creation */
	std::mutex redo_mu;
	struct redo_slot { unsigned *d; size_t cap; hipEvent_t ev; int busy; unsigned *h_left; size_t n_last; } redo[4];
	unsigned redo_next;
	/* Feedback: the general kernel reports (into the slot's pinned word h_left) how many bursts the normal-burst kernel left it.
	 * Read when the slot comes round again (its event has completed by then): a batch that left more than 1/32 of its bursts --
	 * slots of other types the caller gave no hint about -- makes the next 63 eligible launches run the general kernel alone
	 * (every left burst is read twice and stalls the first kernel's prefetch), then one launch probes again. */
	unsigned split_backoff;
	int no_backoff;                    /* TRXHIP_NO_BACKOFF at creation: measurement switch */
};
#define TRX_POOL_SLOTS 1024
#define TRX_REDO_SLOTS 4
#define TRX_REDO_HDR_WORDS 16           /* = TRX_REDO_HDR of the kernels */

/* the TRXD wire packer's launcher (trx_aux_kernels.hip); d_results_copy (may be NULL): every result record is also written
 * there -- the host pipe points it at pinned memory and saves the download */
extern "C" int trx_launch_pack_trxd_wire(const trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
					 const float *d_soft, int soft_stride, const trxhip_trxd_meta *d_meta, uint8_t *d_pkt,
					 int pkt_stride, uint16_t *d_pkt_len, size_t n_bursts, float rssi_offset, hipStream_t stream,
					 trxhip_burst_result *d_results_copy);

/* bursts by reference (trx_aux_kernels.hip): n bursts of `dwords` 4-byte words each from the device-side addresses d_src[] */
extern "C" int trx_launch_gather_bursts(const unsigned long long *d_src, void *d_dst, size_t n, unsigned dwords, hipStream_t stream);

static inline int with_device(const trxhip_ctx *ctx)
{
	return hipSetDevice(ctx->device) == hipSuccess ? 0 : TRXHIP_EIO;
}
#endif
