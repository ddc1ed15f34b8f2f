// trxWrap.h -- the GPU implementations of the sigProcLib.h receive-side API under names of their own (namespace trxgpu),
// for the zero-source-change link recipe: an osmo-trx binary that keeps its own sigProcLib.o
// (the Tx-side modulators, Transceiver.cpp:107-120,392-394, live in the same translation unit as detectAnyBurst) is
// linked with  -Wl,@trxwrap.ldflags libtrxwrap.a -ltrxsigproc -ltrxhip : the linker then routes the callers' references
// to the receive-side functions to libtrxwrap.a's __wrap_ definitions, which call these.
#ifndef TRX_HOST_TRXWRAP_H
#define TRX_HOST_TRXWRAP_H
#include "sigProcLib.h"

namespace trxgpu {
bool sigProcLibSetup();                      /* creates the GPU context only (the tables for the Tx side stay the caller's) */
void sigProcLibDestroy();
int detectAnyBurst(const signalVector &burst, unsigned tsc, float threshold, int sps, CorrType type, unsigned max_toa,
		   struct estim_burst_params *ebp);
SoftVector *demodAnyBurst(const signalVector &burst, CorrType type, int sps, struct estim_burst_params *ebp);
float energyDetect(const signalVector &rxBurst, unsigned windowLength);
void vectorSlicer(float *dest, const float *src, size_t len);
signalVector *delayVector(const signalVector *in, signalVector *out, float delay);
void scaleVector(signalVector &x, complex scale);
int detectSCHBurst(signalVector &burst, float thresh, int sps, sch_detect_type state, struct estim_burst_params *ebp);
}
#endif
