// compat/Vector.h -- STAND-ALONE BUILD ONLY, see compat/Complex.h.
//
// A small aliasing vector with the public surface the receive path of the shim touches on the reference's
// Vector<T> (CommonLibs/Vector.h:56-318): size(), bytes(), begin(), end(), operator[], fill(), clone(), resize(),
// the (size, alloc, free), (block, span) and (data, start, end) constructors.  The object holds the same five
// words in the same order (block, first, last, allocator, deallocator), so sizeof/offsets agree with the
// reference's layout (checked by tests/test_shim_abi.py), but the type lives in trxhip_sa:: and is never handed
// across a library boundary to reference-compiled code.
#ifndef TRXHIP_SA_VECTOR_H
#define TRXHIP_SA_VECTOR_H
#include <cstddef>
#include <cstring>
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
};
TRX_SHIM_NS_END
#endif
