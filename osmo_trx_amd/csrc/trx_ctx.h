// trx_ctx.h -- the context object behind trxhip_ctx* (internal to csrc/; the C ABI only hands out the pointer).
#ifndef TRX_CTX_H
#define TRX_CTX_H
#include <hip/hip_runtime.h>
#include "../../include/trxhip.h"
#include "trx_tables.h"

struct trxhip_ctx {
	int device;
	int n_cu;
	trx_tables *d_tables;
	int no_unit;        /* tables do not have the compiled-in unit structure: keep the multiplying correlation */
	int no_sym;         /* decimator taps not bitwise symmetric: the kernels' straight-line paths (mirrored taps) are off */
	/* cross-die work pool of the 4-SPS kernel (trx_kernel4.hip): one 64-byte counter per launch in flight, handed out
	 * round-robin so that concurrent launches of one context (several host threads, several streams) never share one */
	unsigned *d_pool;
	unsigned pool_next;
};
#define TRX_POOL_SLOTS 64

/* the TRXD wire packer's launcher (trx_aux_kernels.hip); d_results_copy (may be NULL): every result record is also written
 * there -- the host pipe points it at pinned memory and saves the download */
extern "C" int trx_launch_pack_trxd_wire(const trxhip_burst_result *d_results, const trxhip_burst_params *d_params,
					 const float *d_soft, int soft_stride, const trxhip_trxd_meta *d_meta, uint8_t *d_pkt,
					 int pkt_stride, uint16_t *d_pkt_len, size_t n_bursts, float rssi_offset, hipStream_t stream,
					 trxhip_burst_result *d_results_copy);

static inline int with_device(const trxhip_ctx *ctx)
{
	return hipSetDevice(ctx->device) == hipSuccess ? 0 : TRXHIP_EIO;
}
#endif
