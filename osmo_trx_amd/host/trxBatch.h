// trxBatch.h -- what the GPU build adds on top of sigProcLib.h: the batched forms of the pullRadioVector() DSP core
// (Transceiver.cpp:665-815), the TRXD v0/v1 wire packer (proto_trxd.c:28-117) and the burst gatherer that keeps
// pullRadioVector(chan, bi)'s per-burst, per-channel, blocking semantics on top of batched launches.
//
// "sigProcLib.h" is osmo-trx's own header in the product build (libtrxsigproc.so, -I<osmo-trx>/Transceiver52M
// -I<osmo-trx>/CommonLibs) and host/compat/sigProcLib.h in the stand-alone build (libtrxsigproc_sa.so).
#ifndef TRX_HOST_TRXBATCH_H
#define TRX_HOST_TRXBATCH_H
#include <cstddef>
#include <cstdint>
#include "sigProcLib.h"

#ifndef TRX_SHIM_NS_BEGIN              /* built against the reference's headers: its types are global */
#define TRX_SHIM_NS_BEGIN
#define TRX_SHIM_NS_END
#define TRX_SHIM_ABI "reference"
#define TRX_SHIM_REFERENCE_ABI 1
#endif

TRX_SHIM_NS_BEGIN

/** The Viterbi alternative of pullRadioVector (cfg->use_va): demodAnyBurst_va(), a file-static of the reference's
 *  Transceiver.cpp (:620-645) over grgsm_vitac/.  `burst` is the already scaled vector (Transceiver.cpp:783).
 *  Returns a new SoftVector of 156 values (+-127, trailing zeros) the caller deletes, or NULL. */
SoftVector *demodAnyBurst_va(const signalVector &burst, CorrType type, int sps, int rach_max_toa, int tsc);

/* ---- batched form of the pullRadioVector() DSP core (Transceiver.cpp:724-803) ---- */
struct BurstRequest {
	const int16_t *iq;     /* burst_len x (I,Q) as delivered by RadioDevice::readSamples; with diversity (BurstGathererConfig::
	                        * n_paths > 1): the n_paths paths of the burst back to back (radioVector::getVector(i)) */
	CorrType type;         /* expectedCorrType() for the slot */
	unsigned tsc;
	unsigned max_toa;
	uint32_t fn;           /* burstTime.FN(), .TN(): only carried through to the indication / TRXD header */
	uint8_t tn;
};
struct BurstIndication {          /* struct trx_ul_burst_ind (proto_trxd.h:24-37) + the detector's return code */
	float rx_burst[EDGE_BURST_NBITS];     /* soft bits 0..1; nbits of them valid (148 GMSK, 444 8-PSK) */
	unsigned nbits;
	uint32_t fn;
	uint8_t tn;
	double rssi;           /* dBFS incl. rssi_offset */
	double toa;
	bool idle;
	uint8_t modulation;    /* 0 = MODULATION_GMSK, 1 = MODULATION_8PSK (proto_trxd.h:13-21) */
	uint8_t tss;           /* training sequence set: 0 (Transceiver.cpp:703) */
	uint8_t tsc;
	float ci;
	int rc;                /* detectAnyBurst() result, for the rate counters (Transceiver.cpp:769-781) */
	float energy;          /* energyDetect(): avg = sqrt(energy) feeds the caller's noise average (:741-748) */
	uint8_t type;          /* expectedCorrType() of the slot as it was pushed (IDLE slots update the noise ring, :743-748) */
};
/** Process n bursts in one GPU launch.  egprs: some slots may carry 8-PSK (cfg->egprs): soft output is 444 wide.
 *  Host buffers go through the pinned, stream-pipelined path of the C ABI (trxhip_hostpipe_*): no per-call
 *  allocation.  Returns 0, or a negative errno-style code (-EIO on a GPU error). */
int pullRadioVectorBatch(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			 double rssi_offset, BurstIndication *out, bool egprs = false);
/** The same with cfg->use_va (Transceiver.cpp:760-768, :782-784): req[i].iq is the burst read 20 samples early
 *  (osmo-trx.cpp:87-100); power / RSSI come from it, detection runs on the copy shifted by 20 samples, the soft bits
 *  from scaleVector(1/16383) + demodAnyBurst_va() on the unshifted burst.  Three launches on one stream. */
int pullRadioVectorBatchVA(const BurstRequest *req, size_t n, int sps, size_t burst_len, double rxFullScale,
			   double rssi_offset, BurstIndication *out);

/* ---- TRXD uplink burst indications, exactly the bytes trxd_send_burst_ind_v0/v1 write() (proto_trxd.c:68-117) ---- */
#define TRXD_V0_HDR_LEN 8      /* sizeof(struct trxd_hdr_v0): common 5 + v0 3 (proto_trxd.h:56-74) */
#define TRXD_V1_HDR_LEN 11     /* sizeof(struct trxd_hdr_v1): + v1 3 (proto_trxd.h:91-106) */
#define TRXD_MAX_PKT_LEN (TRXD_V1_HDR_LEN + EDGE_BURST_NBITS)
/** Host-side packer for one indication (the reference's arithmetic, restated): returns the datagram length written to
 *  buf (>= TRXD_MAX_PKT_LEN bytes), 0 when nothing is sent (v0 drops idle indications, proto_trxd.c:71-73),
 *  -1 for an unknown version. */
int trxdPackBurstInd(uint8_t *buf, const BurstIndication *bi, unsigned version);

/* ---- the gather stage: N bursts or a timeout, per-channel order, 32-deep drop rule ----
 * Producer side = RadioInterface::driveReceiveRadio() (radioInterface.cpp:272-291: one burst per channel and
 * timeslot, written to mReceiveFIFO[chan] unless 32 are already waiting); consumer side = pullRadioVector(chan, bi)
 * on the RxUpper<chan> thread (Transceiver.cpp:665-815, blocking FIFO read :683).  Between them the gatherer copies
 * each burst straight into the pinned staging buffer of the batch being filled, launches the batch when it holds
 * `max_batch` bursts or its first burst has waited `timeout_us`, and hands the results back per channel in
 * arrival order.  (by_reference: it records the burst's address in a registered receive ring instead, and the device
 * fetches the batch from there.) */
#define TRX_GATHERER_MAX_DEVICES 16
struct BurstGathererConfig {
	size_t chans;             /* number of ARFCN channels (FIFOs) */
	size_t max_batch;         /* launch when this many bursts are gathered (all channels together) */
	unsigned timeout_us;      /* ... or when the oldest gathered burst has waited this long */
	size_t fifo_depth;        /* per-channel bursts pushed and not yet pulled before push() drops: 32 (radioInterface.cpp:277) */
	int sps;                  /* 4 (or 1) */
	size_t burst_len;         /* 625 at 4 SPS */
	double rxFullScale;       /* mRadioInterface->fullScaleOutputValue() */
	double rssi_offset;
	bool egprs;               /* 8-PSK slots possible: 444-bit rows */
	int trxd_version;         /* -1: float soft bits in BurstIndication::rx_burst; 0 / 1: TRXD datagrams packed on the GPU
	                           * (the initial header version of every channel, see setTrxdVersion()) */
	int depth;                /* staging batches in flight (>= 2) */
	int n_paths;              /* diversity paths per burst, radioVector::chans() (0 / 1: none): the path with the highest energy is
	                           * demodulated, rssi and energy come from the path average (Transceiver.cpp:723-751) */
	/* Multi-GPU dispatch (north star: "shard the batch across the 8 GPUs ... host code stays C++").  n_devices = 0: the one
	 * device of sigProcLibSetup() (TRXHIP_DEVICE) -- unless the environment names a list, TRXHIP_DEVICES=0,1,2,3,4,5,6,7.
	 * n_devices >= 1: every entry of devices[] gets its own trxhip_ctx (tables generated once on the host, uploaded per device
	 * with trxhip_create_from_tables: SURVEY section 5 -- no collective needed inside one process), its own host pipe with
	 * `depth` pinned staging batches and its own streams; gathered batches go to the entries round-robin and are delivered
	 * in submission order, so a channel's bursts still come back in push order.  An entry may repeat a device (two
	 * contexts on one GPU: what the single-GPU test runs).  The reference's counterpart is one RxUpper thread per ARFCN
	 * (Transceiver.cpp:1330-1351): here the ARFCNs' bursts share batches and the batches share the GPUs. */
	int n_devices;
	int devices[TRX_GATHERER_MAX_DEVICES];
	bool exact_demod;         /* TRXHIP_FLAG_EXACT_DEMOD: the reference's two FIR stages, soft bits bit-identical to generic C
	                           * (default false: the fused demodulator, include/trxhip.h) */
	bool use_va;              /* cfg->use_va ("viterbi-eq", Transceiver.cpp:760-787): pushed bursts are what the radio read 20
	                           * samples early (osmo-trx.cpp:87-100); power from them as read, detection on the shifted copy, soft
	                           * bits from scaleVector(1/16383) + demodAnyBurst_va() -- all on the GPU, one batch = one stream
	                           * (TRXHIP_FLAG_USE_VA of the host pipe).  Not with exact_demod (there is no filter demodulator). */
	int n_completers;         /* completion threads (0: one per device entry).  Each waits for a batch of its own, sorts it by
	                           * channel and writes the channels' rings when its batch's turn comes (tickets in submission
	                           * order): deliveries stay in push order per channel, two threads overlap the copy of batch k + 1
	                           * with that of batch k.  TRXHIP_COMPLETERS in the environment overrides. */
	bool by_reference;        /* round 5: push() records BurstRequest::iq instead of copying the burst.  The reference cuts each
	                           * burst out of the receive ring with one CPU copy (radioInterface.cpp:272-291); here the ring is
	                           * registered once (registerBuffer(), before start()) and the GPU fetches the gathered bursts
	                           * from it over the link (trxhip_hostpipe_submit_by_ref): the producer's cost per burst is a few
	                           * stores.  Contract: iq is 4-byte aligned, the whole burst lies inside a registered range, and
	                           * the samples stay unchanged until the burst's indication has been pulled (a channel never has
	                           * more than fifo_depth bursts outstanding: a ring of fifo_depth + 1 burst periods per channel
	                           * satisfies it).  A burst outside every registered range fails its batch: -EIO from pull(). */
};
class BurstGatherer {
public:
	explicit BurstGatherer(const BurstGathererConfig &cfg);
	~BurstGatherer();
	bool start();                                   /* needs sigProcLibSetup(); false without a GPU.  After stop(): restarts with empty FIFOs */
	/* Joins the worker threads: batches already submitted are delivered, bursts gathered but not yet submitted are
	 * discarded; every blocked (and every later) pull() on an empty FIFO returns -EIO.  May race with push()/pull(). */
	void stop();
	/* TRXD header version of one channel (the reference negotiates it per channel: mVersionTRXD[chan],
	 * Transceiver.cpp:1238-1250); applies to bursts pushed afterwards.  false in float mode (trxd_version < 0). */
	bool setTrxdVersion(size_t chan, int version);
	/* by_reference: a host range bursts will be pushed from (the radio's receive ring); pinned and mapped into every device
	 * of the list at start(), released at stop() / destruction.  Before start() only; at most 8 ranges. */
	bool registerBuffer(const void *base, size_t bytes);
	/* producer: false = dropped (channel FIFO full, radioInterface.cpp:277-280) or rejected (an EDGE slot on a gatherer
	 * configured without egprs: the 444-bit rows do not exist) */
	bool push(size_t chan, const BurstRequest &req);
	/* producer, one timeslot of driveReceiveRadio() (radioInterface.cpp:272-281: one burst per channel): n bursts for
	 * the channels chans[0..n), gathered with a single reservation.  accepted[k] (optional) = false where the channel's
	 * FIFO was full; returns the number accepted. */
	size_t pushSlot(const size_t *chans, const BurstRequest *reqs, size_t n, bool *accepted = NULL);
	/* consumer, blocking: 0 = *bi filled (maybe bi->idle), -ENOENT = slot OFF (fn/tn filled), -EIO = stopped or GPU
	 * error -- pullRadioVector()'s contract (Transceiver.cpp:658-664).  With trxd_version >= 0, pkt (>= TRXD_MAX_PKT_LEN
	 * bytes) receives the datagram and *pkt_len its length (0 = nothing to send); rx_burst is then left untouched. */
	int pull(size_t chan, BurstIndication *bi, uint8_t *pkt = NULL, size_t *pkt_len = NULL);
	/* counters */
	uint64_t batches() const;
	uint64_t dropped() const;
	uint64_t rejected() const;
	const BurstGathererConfig &config() const;       /* as normalised by the constructor */
	size_t devices() const;                          /* device entries in use (1 without a list) */
	uint64_t batchesOn(size_t entry) const;          /* batches submitted to devices[entry] */
private:
	struct Impl;
	Impl *impl_;
	BurstGatherer(const BurstGatherer &);
	void operator=(const BurstGatherer &);
};
TRX_SHIM_NS_END
#endif
