/*
 * trxarch.h -- the reference's own link-time kernel seam, exported under its own names by libtrxarch.so.
 *
 * osmo-trx selects its arch kernels at link time (ARCH_LA, Transceiver52M/Makefile.common:31-35); every caller --
 * sigProcLib.cpp:365-383, Resampler.cpp:143, Channelizer.cpp:90, Synthesis.cpp:104, radioInterface.cpp:345 -- is
 * compiled against arch/common/{convolve,convert,fft}.h only.  libtrxarch.so provides exactly those symbols with
 * exactly those semantics (HOST pointers, caller-owned interleaved complex float buffers, return values), each
 * call executed on the MI355X as a batch of one over the C ABI of trxhip.h.  It exists so that the reference's
 * remaining CPU-side callers (Resampler, Channelizer, radioInterface) keep linking when sigProcLib.o + libarch.la
 * are replaced; throughput comes from the batched entry points of trxhip.h, not from here.
 *
 * No CPU fallback: without a usable GPU the convolve calls return -1 (the reference's error value), the convert
 * calls leave `out` untouched and report on stderr, init_fft() returns NULL.
 */
#ifndef TRXARCH_H
#define TRXARCH_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- arch/common/convolve.h:4-26 ----
 * y[i] = sum_{k<h_len} x[i + start - (h_len-1) + k] * h[k], i < len; correlation form, no tap flip; x, y interleaved
 * complex float; h complex (imaginary parts ignored by the _real variants).  x must be readable from sample
 * start - (h_len-1) (which may lie in front of x[0]: the callers keep head room there) up to start + len - 1.
 * Returns len, or -1 on a bounds failure (bounds_check(), convolve_base.c:88-105) or a GPU error.
 * The SSE build of the reference dispatches on h_len and sums in a different order (convolve_sse_3.c); here every
 * variant accumulates in the generic-C order of convolve_base.c:28-54, so convolve_* == base_convolve_*. */
void *convolve_h_alloc(size_t num);                     /* convolve_base.c:142-149: tap buffer for `num` complex taps, free() it */
int convolve_real(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len);
int convolve_complex(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len);
int base_convolve_real(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len);
int base_convolve_complex(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len);
void convolve_init(void);                               /* x86/convolve.c:66-91: here it creates the GPU context */

/* ---- arch/common/convert.h:4-13 ---- */
void convert_float_short(short *out, const float *in, float scale, int len);    /* (short)(in * scale), convert_base.c:20-25 */
void convert_short_float(float *out, const short *in, int len);                 /* (float)in, no scaling, convert_base.c:27-31 */
void base_convert_float_short(short *out, const float *in, float scale, int len);
void base_convert_short_float(float *out, const short *in, int len);
void convert_init(void);

/* ---- arch/common/fft.h:6-11 (fft.c:55-114) ----
 * init_fft() fixes the buffers and the fftwf_plan_many_dft() geometry: `istride` transforms of length m, transform t
 * reading in[j*istride + t] and writing (out + ooffset)[k*ostride + t], in complex samples; forward unless `reverse`.
 * cxvec_fft() runs them and returns 0. */
struct fft_hdl;
struct fft_hdl *init_fft(int reverse, int m, int istride, int ostride, float *in, float *out, int ooffset);
void *fft_malloc(size_t size);
void fft_free(void *ptr);
void free_fft(struct fft_hdl *hdl);
int cxvec_fft(struct fft_hdl *hdl);

#ifdef __cplusplus
}
#endif
#endif /* TRXARCH_H */
