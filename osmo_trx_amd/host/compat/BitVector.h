// compat/BitVector.h -- STAND-ALONE BUILD ONLY, see compat/Complex.h.
// SoftVector as the receive path uses it (CommonLibs/BitVector.h:171-231): a Vector<float> of soft decisions,
// -1..+1 out of demodAnyBurst(), 0..1 after vectorSlicer(); bit() slices at 0 (BitVector.h:236-241).
#ifndef TRXHIP_SA_BITVECTOR_H
#define TRXHIP_SA_BITVECTOR_H
#include "Vector.h"

TRX_SHIM_NS_BEGIN
class SoftVector : public Vector<float> {
public:
	SoftVector(size_t n = 0) : Vector<float>(n) {}
	bool bit(size_t k) const { return mStart[k] > 0.0F; }
};
TRX_SHIM_NS_END
#endif
