// sigProcLib.h -- the reference's receive-side sigProcLib interface (Transceiver52M/sigProcLib.h:27-152),
// implemented on the MI355X through the C ABI of include/trxhip.h.
//
// Same names, argument meaning, return values and ownership rules as the reference, so that
// Transceiver::pullRadioVector() (Transceiver.cpp:665-815) links against this instead of
// sigProcLib.o + libarch.la.  The single-burst calls are batch-of-1 wrappers; pullRadioVectorBatch()
// below is the batched form a GPU-aware caller should use.
#ifndef TRX_HOST_SIGPROCLIB_H
#define TRX_HOST_SIGPROCLIB_H
#include <cstddef>
#include <cstdint>
#include <vector>
#include "signalVector.h"

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

/** The Viterbi alternative of pullRadioVector (cfg->use_va): demodAnyBurst_va(), a file-static of the reference's
 *  Transceiver.cpp (:620-645) over grgsm_vitac/.  `burst` is the already scaled vector (Transceiver.cpp:783).
 *  Returns a new SoftVector of 156 values (+-127, trailing zeros) the caller deletes, or NULL. */
SoftVector *demodAnyBurst_va(const signalVector &burst, CorrType type, int sps, int rach_max_toa, int tsc);

/* ---- batched form of the pullRadioVector() DSP core (Transceiver.cpp:724-803) ---- */
struct BurstRequest {
	const int16_t *iq;     /* burst_len x (I,Q) as delivered by RadioDevice::readSamples */
	CorrType type;         /* expectedCorrType() for the slot */
	unsigned tsc;
	unsigned max_toa;
};
struct BurstIndication {          /* the DSP-derived fields of struct trx_ul_burst_ind (proto_trxd.h:24-37) */
	float rx_burst[EDGE_BURST_NBITS];     /* soft bits 0..1; nbits of them valid (148 GMSK, 444 8-PSK) */
	unsigned nbits;
	double rssi;           /* dBFS incl. rssi_offset */
	double toa;
	bool idle;
	uint8_t tsc;
	float ci;
	int rc;                /* detectAnyBurst() result, for the rate counters (Transceiver.cpp:769-781) */
	float energy;
};
/** Process n bursts in one GPU launch.  egprs: some slots may carry 8-PSK (cfg->egprs): soft output is 444 wide.
 *  Returns 0, or a negative errno-style code (-EIO on a GPU error). */
int pullRadioVectorBatch(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			 double rssi_offset, BurstIndication *out, bool egprs = false);
/** The same with cfg->use_va (Transceiver.cpp:760-768, :782-784): req[i].iq is the burst read 20 samples early
 *  (osmo-trx.cpp:87-100); power / RSSI come from it, detection runs on the copy shifted by 20 samples, the soft bits
 *  from scaleVector(1/16383) + demodAnyBurst_va() on the unshifted burst.  Three launches on one stream. */
int pullRadioVectorBatchVA(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			   double rssi_offset, BurstIndication *out);
#endif
