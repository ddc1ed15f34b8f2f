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

static inline int with_device(const trxhip_ctx *ctx)
{
	return hipSetDevice(ctx->device) == hipSuccess ? 0 : TRXHIP_EIO;
}
#endif
