// signalVector.h -- the burst containers at the sigProcLib boundary, host side.
// Same public surface as the reference's Vector<T> / signalVector / SoftVector for what the receive
// path touches (CommonLibs/Vector.h:56-318, Transceiver52M/signalVector.h:13-53, CommonLibs/BitVector.h:171-231):
// size(), begin(), end(), operator[], head-room aware construction, SoftVector::bit().
#ifndef TRX_HOST_SIGNALVECTOR_H
#define TRX_HOST_SIGNALVECTOR_H
#include <cstddef>
#include <cstring>
#include <vector>
#include "Complex.h"

template <class T> class Vector {
public:
	typedef T *iterator;
	typedef const T *const_iterator;
	Vector(size_t n = 0) : store_(n), start_(store_.data()), len_(n) {}
	/* alias an existing block (not owned), like Vector(T* wStart, size_t span) */
	Vector(T *data, size_t n) : start_(data), len_(n) {}
	Vector(const Vector &o) : store_(o.start_, o.start_ + o.len_), start_(store_.data()), len_(o.len_) {}
	Vector &operator=(const Vector &o)
	{
		if (this != &o) { store_.assign(o.start_, o.start_ + o.len_); start_ = store_.data(); len_ = o.len_; }
		return *this;
	}
	size_t size() const { return len_; }
	size_t bytes() const { return len_ * sizeof(T); }
	T *begin() { return start_; }
	const T *begin() const { return start_; }
	T *end() { return start_ + len_; }
	const T *end() const { return start_ + len_; }
	T &operator[](size_t i) { return start_[i]; }
	const T &operator[](size_t i) const { return start_[i]; }
	void fill(const T &v) { for (size_t i = 0; i < len_; i++) start_[i] = v; }
protected:
	std::vector<T> store_;
	T *start_;
	size_t len_;
};

class signalVector : public Vector<complex> {
public:
	signalVector(size_t size = 0) : Vector<complex>(size), real_(false) {}
	/* with head-room, like signalVector(size, start): radioVector(time, 625, head=41) (radioInterface.cpp:266,274) */
	signalVector(size_t size, size_t start) : Vector<complex>(size + start), real_(false)
	{
		start_ += start;
		len_ = size;
		head_ = start;
	}
	/* alias an existing buffer: signalVector(complex *data, size_t start, size_t span) */
	signalVector(complex *data, size_t start, size_t span) : Vector<complex>(data + start, span), real_(false), head_(start) {}
	size_t getStart() const { return head_; }
	bool isReal() const { return real_; }
	void isReal(bool r) { real_ = r; }
private:
	bool real_;
	size_t head_ = 0;
};

/* soft bits: -1..+1 out of demodAnyBurst(), 0..1 after vectorSlicer() */
class SoftVector : public Vector<float> {
public:
	SoftVector(size_t n = 0) : Vector<float>(n) {}
	bool bit(size_t i) const { return (*this)[i] > 0.0F; }      /* BitVector.h:236-241 */
};
#endif
