// compat/sigProcLib.h -- STAND-ALONE BUILD ONLY, see compat/Complex.h.
//
// The receive-side declarations of the reference's Transceiver52M/sigProcLib.h:27-152 (same names, argument
// meaning, return values and ownership rules) for building the shim where no osmo-trx checkout is at hand.
// The product build does NOT use this file: it includes osmo-trx's own sigProcLib.h, so that
// libtrxsigproc.so is link- and layout-compatible with reference-compiled callers (host/README in INTEGRATION.md).
#ifndef TRXHIP_SA_SIGPROCLIB_H
#define TRXHIP_SA_SIGPROCLIB_H
#include <cstddef>
#include <cstdint>
#include "Vector.h"
#include "Complex.h"
#include "BitVector.h"
#include "signalVector.h"

TRX_SHIM_NS_BEGIN

#define NORMAL_BURST_NBITS 148
#define EDGE_BURST_NBITS 444

/** Codes for burst types of received bursts (sigProcLib.h:30-38) */
enum CorrType { OFF, TSC, EXT_RACH, RACH, SCH, EDGE, IDLE };

/** sigProcLib.h:40-46 */
enum SignalError { SIGERR_NONE, SIGERR_BOUNDS, SIGERR_CLIP, SIGERR_UNSUPPORTED, SIGERR_INTERNAL };

#define BURST_THRESH 4.0

/** estimated burst parameters (sigProcLib.h:113-118) */
struct estim_burst_params {
	complex amp;
	float toa;
	uint8_t tsc;
	float ci;
};

/** Setup: generates the tables on the host, uploads them to the GPU, creates the context.
 *  Returns false when no MI355X is usable (there is no CPU fallback).  sigProcLib.h:57 */
bool sigProcLibSetup();
/** sigProcLib.h:60 */
void sigProcLibDestroy(void);

/** Operate soft slicer on a soft-bit vector (sigProcLib.h:63) */
void vectorSlicer(float *dest, const float *src, size_t len);

/** Rough energy estimator (sigProcLib.h:105): mean |x|^2 of windowLength samples taken at stride 4 */
float energyDetect(const signalVector &rxBurst, unsigned windowLength);

/** 8-PSK/GMSK/RACH burst detector (sigProcLib.h:131-137)
 *  @return CorrType (>0) if detected, 0 if not, -SignalError on error */
int detectAnyBurst(const signalVector &burst, unsigned tsc, float threshold, int sps, CorrType type,
		   unsigned max_toa, struct estim_burst_params *ebp);

/** Fractional + integer delay (sigProcLib.h:97, sigProcLib.cpp:1046-1098).  out == NULL: returns a new vector the
 *  caller deletes; otherwise `out` is resized to the result and returned.  NULL on a GPU error. */
signalVector *delayVector(const signalVector *in, signalVector *out, float delay);

/** In-place complex scaling (sigProcLib.h:94, sigProcLib.cpp:1188-1213) */
void scaleVector(signalVector &x, complex scale);

/** SCH synchronisation-burst search of the MS side (sigProcLib.h:139-148, sigProcLib.cpp:1805-1861)
 *  @return 1 if detected (ebp: toa, amp, ci), 0 if not (toa = amp = 0), -1 on error */
enum class sch_detect_type {
	SCH_DETECT_FULL,
	SCH_DETECT_NARROW,
	SCH_DETECT_BUFFER,
};
int detectSCHBurst(signalVector &rxBurst, float detectThreshold, int sps, sch_detect_type state,
		   struct estim_burst_params *ebp);

/** Demodulate burst based on type and output soft bits (sigProcLib.h:151-152).
 *  Returns a new SoftVector the caller deletes (Transceiver.cpp:805), or NULL. */
SoftVector *demodAnyBurst(const signalVector &burst, CorrType type, int sps, struct estim_burst_params *ebp);

TRX_SHIM_NS_END
#endif
