// compat/signalVector.h -- STAND-ALONE BUILD ONLY, see compat/Complex.h.
// The burst container at the sigProcLib boundary with the constructors and accessors the reference's receive
// callers use (Transceiver52M/signalVector.h:13-53): head-room aware construction (radioVector(time, 625, head=41),
// radioInterface.cpp:266,274), aliasing of a caller buffer with the 5-argument form of Transceiver.cpp:680,
// getStart(), isReal(), isAligned().
#ifndef TRXHIP_SA_SIGNALVECTOR_H
#define TRXHIP_SA_SIGNALVECTOR_H
#include "Vector.h"

TRX_SHIM_NS_BEGIN
enum Symmetry { NONE = 0, ABSSYM = 1 };

class signalVector : public Vector<complex> {
public:
	signalVector(size_t size = 0, vector_alloc_func a = NULL, vector_free_func f = NULL)
		: Vector<complex>(size, a, f), real(false), aligned(false), symmetry(NONE) {}
	signalVector(size_t size, size_t start, vector_alloc_func a = NULL, vector_free_func f = NULL)
		: Vector<complex>(size + start, a, f), real(false), aligned(false), symmetry(NONE)
	{
		mStart = mData + start;
	}
	/* existing buffer: the object keeps `data` as its block, exactly as the reference does ("signalvector is owning
	 * despite claiming not to", Transceiver.cpp:648) -- pass a no-op deallocator for memory that is not new[]'ed */
	signalVector(complex *data, size_t start, size_t span, vector_alloc_func a = NULL, vector_free_func f = NULL)
		: Vector<complex>(data, data + start, data + start + span, a, f), real(false), aligned(false), symmetry(NONE) {}
	signalVector(const signalVector &o) : Vector<complex>(o.size() + o.getStart()), aligned(false)
	{
		mStart = mData + o.getStart();
		for (size_t k = 0; k < o.size(); k++)
			mStart[k] = o.mStart[k];
		symmetry = o.symmetry;
		real = o.real;
	}
	void operator=(const signalVector &o)
	{
		const size_t head = o.getStart();
		resize(o.size() + head);
		for (size_t k = 0; k < o.size() + head; k++)
			mData[k] = o.mData[k];
		mStart = mData + head;
	}
	size_t getStart() const { return mStart - mData; }
	Symmetry getSymmetry() const { return symmetry; }
	void setSymmetry(Symmetry s) { symmetry = s; }
	bool isReal() const { return real; }
	void isReal(bool r) { real = r; }
	bool isAligned() const { return aligned; }
	void setAligned(bool a) { aligned = a; }

private:
	bool real;
	bool aligned;
	Symmetry symmetry;
};
TRX_SHIM_NS_END
#endif
