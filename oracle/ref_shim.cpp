/*
 * ref_shim.cpp -- C-callable handle around the reference's own C++ Resampler class
 * (Transceiver52M/Resampler.{h,cpp}, compiled unmodified from /root/reference by oracle/Makefile).
 * TEST INFRASTRUCTURE ONLY: lets tests/ drive the real reference object through ctypes.
 * This file contains no reference code; it only calls the reference's public interface
 * (Resampler.h:27-62).
 */
#include <cstddef>
#include "Resampler.h"

extern "C" {

void *ref_resampler_new(size_t p, size_t q, size_t filt_len, float bw)
{
	Resampler *r = new Resampler(p, q, filt_len);
	if (!r->init(bw)) {
		delete r;
		return nullptr;
	}
	return r;
}

void ref_resampler_free(void *h)
{
	delete static_cast<Resampler *>(h);
}

/* `in` must point filt_len samples into a buffer (history in front), as the reference's callers do */
int ref_resampler_rotate(void *h, const float *in, size_t in_len, float *out, size_t out_len)
{
	return static_cast<Resampler *>(h)->rotate(in, in_len, out, out_len);
}

}
