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

/* soft bits per burst: GMSK, 8-PSK (sigProcLib.h:27-28) */
enum { NORMAL_BURST_NBITS = 148, EDGE_BURST_NBITS = 444 };

/* what a timeslot is expected to carry (sigProcLib.h:30-38); the values are part of the interface (0 .. 6) */
enum CorrType {
	OFF = 0,        /* timeslot is off */
	TSC = 1,        /* normal burst, GMSK */
	EXT_RACH = 2,   /* access burst, training sequences TS0 .. TS2 */
	RACH = 3,       /* access burst, TS0 */
	SCH = 4,        /* synchronisation burst (MS side) */
	EDGE = 5,       /* normal burst, 8-PSK, falls back to TSC */
	IDLE = 6        /* nothing expected: noise measurement */
};

/* negated, these are detectAnyBurst()'s error returns (sigProcLib.h:40-46) */
enum SignalError {
	SIGERR_NONE = 0,
	SIGERR_BOUNDS = 1,
	SIGERR_CLIP = 2,
	SIGERR_UNSUPPORTED = 3,
	SIGERR_INTERNAL = 4
};

/* peak-to-average ratio a correlation has to exceed (sigProcLib.h:48) */
#define BURST_THRESH 4.0

/* the detector's findings, consumed by the demodulator (sigProcLib.h:113-118): 20 bytes, toa at 8, tsc at 12, ci at 16 */
struct estim_burst_params {
	complex amp;    /* channel amplitude */
	float toa;      /* symbols, relative to the expected position */
	uint8_t tsc;    /* training sequence found */
	float ci;       /* dB */
};

/* which part of the buffer detectSCHBurst() searches (sigProcLib.h:139-143) */
enum class sch_detect_type { SCH_DETECT_FULL, SCH_DETECT_NARROW, SCH_DETECT_BUFFER };

/* ---- life cycle (sigProcLib.h:57, :60).  Setup generates the tables on the host, uploads them and creates the GPU
 * context; false when no MI355X is usable -- there is no CPU fallback. */
bool sigProcLibSetup();
void sigProcLibDestroy(void);

/* ---- the receive path in the order pullRadioVector() walks it (Transceiver.cpp:724-803) */

/* mean |x|^2 over `window` samples taken at stride 4 (sigProcLib.h:105) */
float energyDetect(const signalVector &burst, unsigned window);

/* > 0: the CorrType found, 0: nothing, < 0: -SignalError (sigProcLib.h:131-137) */
int detectAnyBurst(const signalVector &burst,
		   unsigned tsc,
		   float threshold,
		   int sps,
		   CorrType expected,
		   unsigned max_toa,
		   struct estim_burst_params *found);

/* soft bits of a detected burst as a new SoftVector the caller deletes (Transceiver.cpp:805), NULL on error
 * (sigProcLib.h:151-152) */
SoftVector *demodAnyBurst(const signalVector &burst, CorrType detected, int sps, struct estim_burst_params *found);

/* -1 .. +1 soft values to 0 .. 1 (sigProcLib.h:63) */
void vectorSlicer(float *out, const float *in, size_t n);

/* ---- helpers that are entry points of their own */

/* fractional + integer delay (sigProcLib.h:97, sigProcLib.cpp:1046-1098).  dst == NULL: a new vector the caller deletes;
 * otherwise dst is resized to the result and returned.  NULL on a GPU error. */
signalVector *delayVector(const signalVector *src, signalVector *dst, float delay);

/* x *= factor, in place (sigProcLib.h:94, sigProcLib.cpp:1188-1213) */
void scaleVector(signalVector &x, complex factor);

/* 1: found (toa, amp, ci filled), 0: not found (toa = amp = 0), -1: error (sigProcLib.h:144-148, sigProcLib.cpp:1805-1861) */
int detectSCHBurst(signalVector &buffer, float threshold, int sps, sch_detect_type where, struct estim_burst_params *found);

TRX_SHIM_NS_END
#endif
