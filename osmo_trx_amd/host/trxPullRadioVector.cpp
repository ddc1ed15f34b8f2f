// trxPullRadioVector.cpp -- see trxPullRadioVector.h.  Line references: Transceiver52M/Transceiver.cpp.
#include <cerrno>
#include <cmath>
#include <cstddef>
#include <cstring>

#include "trxPullRadioVector.h"

TRX_SHIM_NS_BEGIN

RxChanState::RxChanState() : noise_itr(0), mNoiseLev(0.0f), mMuted(false), ctr_changed(false), use_rssi_offset(false), rssi_offset(0.0)
{
	for (size_t i = 0; i < TRX_NOISE_CNT; i++)
		noises[i] = 0.0f;                                          /* std::vector<float>(size): value-initialised */
	ctrs.rx_empty_burst = ctrs.rx_clipping = ctrs.rx_no_burst_detected = 0;
}

/* radioVector.cpp:97-108 */
bool RxChanState::insertNoise(float val)
{
	if (noise_itr >= TRX_NOISE_CNT)
		noise_itr = 0;
	noises[noise_itr++] = val;
	return true;
}

/* radioVector.cpp:84-95: float accumulation in index order, then / (float) size() */
float RxChanState::avgNoise() const
{
	float val = 0.0;
	for (size_t i = 0; i < TRX_NOISE_CNT; i++)
		val += noises[i];
	return val / (float)TRX_NOISE_CNT;
}

int trxPullRadioVector(BurstGatherer &g, RxChanState &st, size_t chan, struct trx_ul_burst_ind *bi)
{
	if (!bi)
		return -EIO;
	const BurstGathererConfig &cfg = g.config();
	if (cfg.trxd_version >= 0)
		return -EIO;                                               /* datagram mode delivers no float soft bits */
	BurstIndication ind;
	const int code = g.pull(chan, &ind);                               /* blocking, as mReceiveFIFO[chan]->read() (:683) */
	if (code == -EIO)
		return -EIO;                                               /* :684-687 */

	/* Initialize struct bi (:693-704) */
	bi->nbits = 0;
	bi->fn = ind.fn;
	bi->tn = ind.tn;
	bi->rssi = 0.0;
	bi->toa = 0.0;
	bi->noise = 0.0;
	bi->idle = false;
	bi->modulation = MODULATION_GMSK;
	bi->tss = 0;
	bi->tsc = 0;
	bi->ci = 0.0;

	if (code == -ENOENT)                                               /* type == OFF: not even power or noise (:713-717) */
		return -ENOENT;
	if (st.mMuted) {                                                   /* :719-721 -> ret_idle */
		bi->idle = true;
		return 0;
	}
	/* Diversity paths were compared on the GPU (trxhip_select_diversity_batch); ind.energy = sum_i pow_i / chans = avg^2.
	 * (:723-741; "Received empty burst" -- a radioVector without paths -- cannot reach the gatherer: push() refuses it.) */
	const float avg = sqrtf(ind.energy);                               /* avg = sqrt(avg / radio_burst->chans()) (:741) */
	const bool is_idle_slot = ind.type == IDLE;
	if (is_idle_slot) {                                                /* type == IDLE: update noise levels (:743-748) */
		st.insertNoise(avg);
		st.mNoiseLev = st.avgNoise();
	}
	/* :750-752, in double as there (rxFullScale double, avg / mNoiseLev float) */
	const double rssi_offset = st.use_rssi_offset ? st.rssi_offset : cfg.rssi_offset;   /* rssiOffset(chan), :613-618 */
	bi->rssi = 20.0 * log10(cfg.rxFullScale / avg) + rssi_offset;
	bi->noise = 20.0 * log10(cfg.rxFullScale / st.mNoiseLev) + rssi_offset;
	if (is_idle_slot) {                                                /* :754-755 */
		bi->idle = true;
		return 0;
	}
	if (ind.rc <= 0) {                                                 /* :769-781 */
		if (ind.rc == -SIGERR_CLIP) {
			st.ctrs.rx_clipping++;
			st.ctr_changed = true;
		} else if (ind.rc != SIGERR_NONE) {
			st.ctrs.rx_no_burst_detected++;
			st.ctr_changed = true;
		}
		bi->idle = true;
		return 0;
	}
	bi->toa = ind.toa;                                                 /* :789-791 */
	bi->tsc = ind.tsc;
	bi->ci = ind.ci;
	if (ind.nbits == EDGE_BURST_NBITS) {                               /* :794-800 */
		bi->modulation = MODULATION_8PSK;
		bi->nbits = EDGE_BURST_NBITS;
	} else {
		bi->modulation = MODULATION_GMSK;
		bi->nbits = NORMAL_BURST_NBITS;                            /* gSlotLen */
	}
	memcpy(bi->rx_burst, ind.rx_burst, bi->nbits * sizeof(float));     /* vectorSlicer() ran on the GPU (:803) */
	return 0;
}

TRX_SHIM_NS_END

extern "C" void trxsigproc_abi_layout_bi(size_t out[14])
{
	out[0] = sizeof(struct trx_ul_burst_ind);
	out[1] = offsetof(struct trx_ul_burst_ind, rx_burst);
	out[2] = offsetof(struct trx_ul_burst_ind, nbits);
	out[3] = offsetof(struct trx_ul_burst_ind, fn);
	out[4] = offsetof(struct trx_ul_burst_ind, tn);
	out[5] = offsetof(struct trx_ul_burst_ind, rssi);
	out[6] = offsetof(struct trx_ul_burst_ind, toa);
	out[7] = offsetof(struct trx_ul_burst_ind, noise);
	out[8] = offsetof(struct trx_ul_burst_ind, idle);
	out[9] = offsetof(struct trx_ul_burst_ind, modulation);
	out[10] = offsetof(struct trx_ul_burst_ind, tss);
	out[11] = offsetof(struct trx_ul_burst_ind, tsc);
	out[12] = offsetof(struct trx_ul_burst_ind, ci);
	out[13] = sizeof(enum Modulation);
}
