// compat/Complex.h -- STAND-ALONE BUILD ONLY (no osmo-trx checkout at hand).
//
// The product shim (libtrxsigproc.so) is compiled against osmo-trx's own Transceiver52M/Complex.h; this
// header only exists so that the shim and its selftest can also be built and run where those headers are
// absent (libtrxsigproc_sa.so).  Everything lives in the inline namespace trxhip_sa, so the mangled names of
// the stand-alone build can never collide with (or be mistaken for) the reference's types.
// Same member names as the reference's call sites use (Complex.h:31-118: r, i, real(), imag(), norm2(), abs(),
// conj(); typedef `complex`).  Arithmetic on bursts happens on the GPU, not here.
#ifndef TRXHIP_SA_COMPLEX_H
#define TRXHIP_SA_COMPLEX_H
#include <cmath>

#define TRX_SHIM_NS_BEGIN inline namespace trxhip_sa {
#define TRX_SHIM_NS_END }
#define TRX_SHIM_ABI "standalone"

TRX_SHIM_NS_BEGIN
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
	bool operator!=(const Complex &a) const { return !(*this == a); }
};
typedef Complex<float> complex;
TRX_SHIM_NS_END
#endif
