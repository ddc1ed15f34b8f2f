// shim_internal.h -- shared between the translation units of the host shim; not installed.
#ifndef TRX_HOST_SHIM_INTERNAL_H
#define TRX_HOST_SHIM_INTERNAL_H
#include "trxBatch.h"
#include "trxhip.h"

/* the context sigProcLibSetup() created (sigProcLib.cpp) */
extern "C" trxhip_ctx *trxsigproc_context(void);
/* a further context on `device` with the SAME tables (generated once per process, uploaded with
 * trxhip_create_from_tables): the multi-device gatherer owns one per devices[] entry and destroys it itself */
extern "C" trxhip_ctx *trxsigproc_create_context(int device);
extern "C" void trxsigproc_destroy_context(trxhip_ctx *ctx);

TRX_SHIM_NS_BEGIN
/* result record + soft row of the C ABI -> the fields pullRadioVector() fills in struct trx_ul_burst_ind
 * (Transceiver.cpp:694-704, :751, :789-803) */
void trxsigproc_fill_indication(BurstIndication &bi, const BurstRequest &rq, const trxhip_burst_result &r, const float *soft,
				size_t stride, double rssi_offset);
TRX_SHIM_NS_END
#endif
