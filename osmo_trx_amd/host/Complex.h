// Complex.h -- minimal complex<float> with the member names the reference's call sites use
// (Transceiver52M/Complex.h: real(), imag(), norm2(), abs(), conj(); typedef `complex`).
// Written for the host shim; arithmetic on bursts happens on the GPU, not here.
#ifndef TRX_HOST_COMPLEX_H
#define TRX_HOST_COMPLEX_H
#include <cmath>

template <class Real> class Complex {
public:
	Real r, i;
	Complex() : r(0), i(0) {}
	Complex(Real re) : r(re), i(0) {}
	Complex(Real re, Real im) : r(re), i(im) {}
	Real real() const { return r; }
	Real imag() const { return i; }
	Real norm2() const { return i * i + r * r; }
	Real abs() const { return std::sqrt(norm2()); }
	Complex conj() const { return Complex(r, -i); }
	Complex operator+(const Complex &a) const { return Complex(r + a.r, i + a.i); }
	Complex operator-(const Complex &a) const { return Complex(r - a.r, i - a.i); }
	Complex operator*(const Complex &a) const { return Complex(r * a.r - i * a.i, r * a.i + i * a.r); }
	Complex operator*(Real a) const { return Complex(r * a, i * a); }
	bool operator==(const Complex &a) const { return r == a.r && i == a.i; }
};
typedef Complex<float> complex;
#endif
