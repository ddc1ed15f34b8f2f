// trxPullRadioVector.h -- Transceiver::pullRadioVector(size_t chan, struct trx_ul_burst_ind *bi) (Transceiver.h:205,
// Transceiver.cpp:665-815) over the batched GPU path: same output type, same return codes, same per-channel state.
//
// The reference's function reads one burst from mReceiveFIFO[chan], runs the DSP and fills `bi`.  Here the DSP of all
// channels runs in batches behind a BurstGatherer (trxBatch.h) -- fed by the RxLower side with BurstGatherer::pushSlot()
// where the reference writes the FIFO -- and this function is what the RxUpper<chan> thread calls instead:
//
//     int Transceiver::pullRadioVector(size_t chan, struct trx_ul_burst_ind *bi)
//     { return trxPullRadioVector(*mGatherer, mRxState[chan], chan, bi); }
//
// It blocks as the FIFO read does, fills every field the reference fills (incl. `noise` from the 20-entry noise ring of
// the channel, `modulation` as enum Modulation, `idle`) and returns 0 / -ENOENT (slot OFF: fn and tn filled) / -EIO.
//
// `struct trx_ul_burst_ind` is osmo-trx's own (proto_trxd.h:24-37) when this file is compiled inside an osmo-trx build
// (reference headers on the include path AND libosmocore installed: proto_trxd.h includes <osmocom/core/endian.h>);
// otherwise the field-for-field declaration below.  tests/test_shim_abi.py compares the two field lists and the layout
// on both sides of the library boundary.
#ifndef TRX_HOST_PULLRADIOVECTOR_H
#define TRX_HOST_PULLRADIOVECTOR_H
#include "trxBatch.h"

#if defined(TRX_SHIM_REFERENCE_ABI) && defined(__has_include)
#if __has_include(<osmocom/core/endian.h>) && __has_include("proto_trxd.h")
#include "proto_trxd.h"
#define TRX_HAVE_REFERENCE_PROTO_TRXD 1
#endif
#endif

#ifndef TRX_HAVE_REFERENCE_PROTO_TRXD
/* proto_trxd.h:12-37, declaration only (C linkage types, global namespace as there) */
#ifndef MAX_RX_BURST_BUF_SIZE
#define MAX_RX_BURST_BUF_SIZE 444 /* 444 = EDGE_BURST_NBITS */
enum Modulation {
	MODULATION_GMSK,
	MODULATION_8PSK,
};
struct trx_ul_burst_ind {
	float rx_burst[MAX_RX_BURST_BUF_SIZE]; /* soft bits normalized 0..1 */
	unsigned nbits;  /* number of symbols per slot in rxBurst, not counting guard periods */
	uint32_t fn;     /* TDMA frame number */
	uint8_t tn;      /* TDMA time-slot number */
	double rssi;     /* in dBFS */
	double toa;      /* in symbols */
	double noise;    /* noise level in dBFS */
	bool idle;       /* true if no valid burst is included */
	enum Modulation modulation;
	uint8_t tss;     /* training sequence set */
	uint8_t tsc;     /* training sequence code */
	float ci;        /* Carrier-to-Interference ratio, in dB */
};
#endif
#endif

TRX_SHIM_NS_BEGIN

#define TRX_NOISE_CNT 20               /* NOISE_CNT, Transceiver.cpp:55 */

/** The receive-side members of TransceiverState (Transceiver.h:56-101) that pullRadioVector() reads and writes, per channel:
 *  the noise ring (avgVector mNoises(NOISE_CNT), radioVector.cpp:84-108), mNoiseLev, mMuted and the three rate counters it
 *  bumps (struct trx_counters, osmo_signal.h:68-70).  One object per channel, touched by that channel's RxUpper thread only. */
struct RxChanState {
	float noises[TRX_NOISE_CNT];       /* avgVector: zero-initialised, overwritten round-robin */
	size_t noise_itr;
	float mNoiseLev;                   /* 0.0 until the first IDLE slot (Transceiver.cpp:66): noise = +inf dB until then, as in the reference */
	bool mMuted;                       /* TRXC "MUTE": idle indications, no power measurement (:719-721) */
	struct { unsigned rx_empty_burst, rx_clipping, rx_no_burst_detected; } ctrs;
	bool ctr_changed;                  /* set when a counter moved: the caller dispatches its rate-counter signal (:806-807) and clears it */
	/* Transceiver::rssiOffset(chan) = mRadioInterface->rssiOffset(chan) + cfg->rssi_offset, per channel unless
	 * force_rssi_offset is set (Transceiver.cpp:613-618, :750-752): set use_rssi_offset and rssi_offset to that sum for a
	 * channel whose RX gain differs; otherwise the gatherer-wide BurstGathererConfig::rssi_offset applies */
	bool use_rssi_offset;
	double rssi_offset;
	RxChanState();
	bool insertNoise(float val);       /* avgVector::insert() */
	float avgNoise() const;            /* avgVector::avg(): serial float sum over all 20 entries / 20 */
};

/** pullRadioVector(chan, bi).  Blocks until the channel's next burst has come back from the GPU.
 *  0: *bi filled (bi->idle for IDLE slots, misses and muted channels); -ENOENT: the slot is OFF (bi->fn / bi->tn filled,
 *  no power or noise update); -EIO: gatherer stopped or GPU error.  The gatherer must run in float mode
 *  (BurstGathererConfig::trxd_version = -1); rxFullScale is the gatherer's, rssi_offset the channel's (RxChanState) or the gatherer's. */
int trxPullRadioVector(BurstGatherer &g, RxChanState &st, size_t chan, struct trx_ul_burst_ind *bi);

TRX_SHIM_NS_END

/* sizeof / offsetof of struct trx_ul_burst_ind as THIS library was compiled (tests/test_shim_abi.py) */
extern "C" void trxsigproc_abi_layout_bi(size_t out[14]);
#endif
