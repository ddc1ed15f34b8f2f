// MultiArfcnRx.cpp -- see MultiArfcnRx.h.  All DSP runs on the GPU through trxhip_rx_frontend_* (include/trxhip.h).
#include <hip/hip_runtime.h>

#include <cerrno>

#include "MultiArfcnRx.h"
#include "trxhip.h"

extern "C" trxhip_ctx *trxsigproc_context(void);      /* sigProcLib.cpp: the context sigProcLibSetup() created */

MultiArfcnRx::MultiArfcnRx(size_t chans, size_t block_len, int resamp_p, int resamp_q)
	: chans_(chans), block_len_(block_len), p_(resamp_p), q_(resamp_q), fe_(nullptr), stream_(nullptr),
	  d_wide_(nullptr), d_out_(nullptr), cap_blocks_(0)
{
}

MultiArfcnRx::~MultiArfcnRx()
{
	if (fe_) trxhip_rx_frontend_destroy(fe_);
	if (d_wide_) hipFree(d_wide_);
	if (d_out_) hipFree(d_out_);
	if (stream_) hipStreamDestroy(static_cast<hipStream_t>(stream_));
}

/* radioInterfaceMulti.cpp:87-122 */
int MultiArfcnRx::getLogicalChan(size_t pchan, size_t chans)
{
	switch (chans) {
	case 1: return pchan == 0 ? 0 : -1;
	case 2: return pchan == 0 ? 0 : (pchan == 3 ? 1 : -1);
	case 3: return pchan == 1 ? 0 : (pchan == 0 ? 1 : (pchan == 3 ? 2 : -1));
	default: return -1;
	}
}

bool MultiArfcnRx::init()
{
	if (chans_ < 1 || chans_ > 3 || !trxsigproc_context())
		return false;
	hipStream_t s;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
		return false;
	stream_ = s;
	return trxhip_rx_frontend_create(trxsigproc_context(), (int)block_len_, p_, q_, &fe_) == TRXHIP_OK;
}

int MultiArfcnRx::pullBuffer(const int16_t *wide, size_t n_blocks, std::vector<std::vector<complex> > &out)
{
	if (!fe_ || !wide)
		return -EIO;
	if (!n_blocks)
		return 0;
	const size_t n_wide = n_blocks * block_len_ * MCHANS;           /* complex int16 samples */
	const size_t n_out = n_blocks * block_len_ / q_ * p_;           /* per channel */
	hipStream_t s = static_cast<hipStream_t>(stream_);
	if (n_blocks > cap_blocks_) {
		if (d_wide_) hipFree(d_wide_);
		if (d_out_) hipFree(d_out_);
		d_wide_ = d_out_ = nullptr;
		if (hipMalloc(&d_wide_, n_wide * 4) != hipSuccess || hipMalloc(&d_out_, MCHANS * n_out * 8) != hipSuccess) {
			cap_blocks_ = 0;
			return -EIO;
		}
		cap_blocks_ = n_blocks;
	}
	if (hipMemcpyAsync(d_wide_, wide, n_wide * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
	    trxhip_rx_frontend_pull(fe_, static_cast<const int16_t *>(d_wide_), n_blocks, static_cast<float *>(d_out_), n_out, s) != TRXHIP_OK)
		return -EIO;
	out.resize(chans_);
	for (size_t pchan = 0; pchan < MCHANS; pchan++) {
		const int lchan = getLogicalChan(pchan, chans_);
		if (lchan < 0)
			continue;
		std::vector<complex> &dst = out[lchan];
		const size_t old = dst.size();
		dst.resize(old + n_out);
		if (hipMemcpyAsync(&dst[old], static_cast<const char *>(d_out_) + pchan * n_out * 8, n_out * 8,
				   hipMemcpyDeviceToHost, s) != hipSuccess)
			return -EIO;
	}
	return hipStreamSynchronize(s) == hipSuccess ? 0 : -EIO;
}
