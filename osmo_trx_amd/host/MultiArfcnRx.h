// MultiArfcnRx.h -- the receive half of RadioInterfaceMulti (Transceiver52M/radioInterfaceMulti.{h,cpp}) on the GPU:
// wideband int16 chunks in, per-ARFCN 4-SPS sample streams out.  Same constants and channel mapping as the
// reference: MCHANS = 4 filterbank paths, 192-sample channelizer blocks, Resampler(65, 48) to the GSM rate
// (radioInterfaceMulti.cpp:35-42), physical -> logical channel map getLogicalChan() (:87-122).
#ifndef TRX_HOST_MULTIARFCNRX_H
#define TRX_HOST_MULTIARFCNRX_H
#include <cstddef>
#include <cstdint>
#include <vector>
#include "signalVector.h"

struct trxhip_rx_frontend;

class MultiArfcnRx {
public:
	static const size_t MCHANS = 4;                 /* radioInterfaceMulti.cpp:42 */
	explicit MultiArfcnRx(size_t chans, size_t block_len = 192, int resamp_p = 65, int resamp_q = 48);
	~MultiArfcnRx();
	bool init();                                    /* needs sigProcLibSetup() first; false without a GPU */
	/* One pullBuffer() worth of work for n_blocks channelizer blocks (radioInterfaceMulti.cpp:237-314):
	 * wide = n_blocks * block_len * MCHANS int16 IQ samples as read from the device.  Appends
	 * n_blocks * block_len * p / q samples to out[lchan] for every logical channel.  0 or -EIO. */
	int pullBuffer(const int16_t *wide, size_t n_blocks, std::vector<std::vector<complex> > &out);
	size_t chans() const { return chans_; }
	/* radioInterfaceMulti.cpp:87-122 */
	static int getLogicalChan(size_t pchan, size_t chans);
private:
	size_t chans_, block_len_;
	int p_, q_;
	trxhip_rx_frontend *fe_;
	void *stream_, *d_wide_, *d_out_;
	size_t cap_blocks_;
};
#endif
