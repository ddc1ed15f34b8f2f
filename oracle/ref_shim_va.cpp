// ref_shim_va.cpp -- C handle around the reference's own viterbi_detector() (Transceiver52M/grgsm_vitac/
// viterbi_detector.cc, compiled unmodified from /root/reference by oracle/Makefile into oracle/_ref/libref_va.so).
// Test infrastructure: pins oracle/trx_oracle.c:orc_va_viterbi().  grgsm_vitac.cpp itself needs libosmocore's
// <osmocom/core/bits.h> and is not buildable here.
#include "viterbi_detector.h"

extern "C" void ref_viterbi_detector(const float *in, unsigned n, const float *rhh, unsigned start_state,
				     const unsigned *stop_states, unsigned nstops, float *out)
{
	viterbi_detector(reinterpret_cast<const gr_complex *>(in), n,
			 const_cast<gr_complex *>(reinterpret_cast<const gr_complex *>(rhh)), start_state, stop_states, nstops, out);
}
