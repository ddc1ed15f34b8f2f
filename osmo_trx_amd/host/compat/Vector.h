// compat/Vector.h -- STAND-ALONE BUILD ONLY, see compat/Complex.h.
//
// A small aliasing vector with the public surface the receive path of the shim touches on the reference's
// Vector<T> (CommonLibs/Vector.h:56-318): size(), bytes(), begin(), end(), operator[], fill(), clone(), resize(),
// the (size, alloc, free), (block, span) and (data, start, end) constructors, plus what the reference's own
// tests/CommonLibs/VectorTest.cpp exercises (concatenating constructor, segment()/head()/tail() aliases,
// copyTo()/copyToSegment(), operator<<): that test compiles unmodified against this header and reproduces
// VectorTest.ok (tests/test_shim_abi.py).  The object holds the same five
// words in the same order (block, first, last, allocator, deallocator), so sizeof/offsets agree with the
// reference's layout (checked by tests/test_shim_abi.py), but the type lives in trxhip_sa:: and is never handed
// across a library boundary to reference-compiled code.
#ifndef TRXHIP_SA_VECTOR_H
#define TRXHIP_SA_VECTOR_H
#include <cassert>
#include <cstddef>
#include <cstring>
#include <ostream>
#include "Complex.h"

TRX_SHIM_NS_BEGIN
typedef void (*vector_free_func)(void *wData);
typedef void *(*vector_alloc_func)(size_t newSize);

template <class T> class Vector {
public:
	typedef T *iterator;
	typedef const T *const_iterator;

protected:
	T *mData;   /* owned block or NULL (alias) */
	T *mStart;  /* first useful element */
	T *mEnd;    /* one past the last */
	vector_alloc_func mAllocFunc;
	vector_free_func mFreeFunc;

	void release()
	{
		if (!mData)
			return;
		if (mFreeFunc)
			mFreeFunc(mData);
		else
			delete[] mData;
		mData = NULL;
	}

public:
	Vector(size_t n = 0, vector_alloc_func a = NULL, vector_free_func f = NULL) : mData(NULL), mAllocFunc(a), mFreeFunc(f)
	{
		resize(n);
	}
	Vector(T *data, T *start, T *end, vector_alloc_func a = NULL, vector_free_func f = NULL)
		: mData(data), mStart(start), mEnd(end), mAllocFunc(a), mFreeFunc(f) {}
	/* alias of an existing block, never freed */
	Vector(T *start, size_t span, vector_alloc_func a = NULL, vector_free_func f = NULL)
		: mData(NULL), mStart(start), mEnd(start + span), mAllocFunc(a), mFreeFunc(f) {}
	Vector(const Vector &o) : mData(NULL), mAllocFunc(o.mAllocFunc), mFreeFunc(o.mFreeFunc) { clone(o); }
	/* a followed by b, in a block of its own (Vector.h:163-170) */
	Vector(const Vector &a, const Vector &b, vector_alloc_func al = NULL, vector_free_func fr = NULL)
		: mData(NULL), mAllocFunc(al), mFreeFunc(fr)
	{
		resize(a.size() + b.size());
		a.copyTo(*this);
		b.copyToSegment(*this, a.size());
	}
	~Vector() { release(); }
	void operator=(const Vector &o) { clone(o); }

	/* new size, content discarded */
	void resize(size_t n)
	{
		release();
		if (n)
			mData = mAllocFunc ? static_cast<T *>(mAllocFunc(n)) : new T[n];
		mStart = mData;
		mEnd = mStart + n;
	}
	void clone(const Vector &o)
	{
		resize(o.size());
		for (size_t k = 0; k < o.size(); k++)
			mStart[k] = o.mStart[k];
	}
	size_t size() const { return mEnd - mStart; }
	size_t bytes() const { return size() * sizeof(T); }
	T *begin() { return mStart; }
	const T *begin() const { return mStart; }
	T *end() { return mEnd; }
	const T *end() const { return mEnd; }
	T &operator[](size_t k) { return mStart[k]; }
	const T &operator[](size_t k) const { return mStart[k]; }
	void fill(const T &v)
	{
		for (T *p = mStart; p < mEnd; p++)
			*p = v;
	}
	bool isOwner() { return mData != NULL; }

	/* aliases into this vector's block: they own nothing and die with it (Vector.h:202-224) */
	Vector segment(size_t first, size_t span)
	{
		assert(mStart + first + span <= mEnd);
		return Vector(NULL, mStart + first, mStart + first + span);
	}
	const Vector segment(size_t first, size_t span) const
	{
		assert(mStart + first + span <= mEnd);
		return Vector(NULL, mStart + first, mStart + first + span);
	}
	Vector head(size_t span) { return segment(0, span); }
	const Vector head(size_t span) const { return segment(0, span); }
	Vector tail(size_t first) { return segment(first, size() - first); }
	const Vector tail(size_t first) const { return segment(first, size() - first); }

	/* element copies into a vector that already has room (Vector.h:226-262) */
	void copyToSegment(Vector &dst, size_t first, size_t span) const
	{
		assert(dst.mStart + first + span <= dst.mEnd && mStart + span <= mEnd);
		for (size_t k = 0; k < span; k++)
			dst.mStart[first + k] = mStart[k];
	}
	void copyToSegment(Vector &dst, size_t first = 0) const { copyToSegment(dst, first, size()); }
	void copyTo(Vector &dst) const { copyToSegment(dst, 0, size()); }
	void segmentCopyTo(Vector &dst, size_t first, size_t span) const { segment(first, span).copyTo(dst); }
};

/* every element followed by a blank (Vector.h:324-329) */
template <class T> std::ostream &operator<<(std::ostream &os, const Vector<T> &v)
{
	for (size_t k = 0; k < v.size(); k++)
		os << v[k] << " ";
	return os;
}
TRX_SHIM_NS_END
#endif
