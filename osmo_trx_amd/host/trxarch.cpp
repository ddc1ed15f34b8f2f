// trxarch.cpp -- libtrxarch.so: the reference's arch seam (include/trxarch.h = arch/common/{convolve,convert,fft}.h)
// with host-pointer semantics, every call a batch of one over the C ABI of trxhip.h.  No arithmetic happens here:
// the window of x the kernel needs goes up, the result comes back.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "trxarch.h"
#include "trxhip.h"

namespace {

trxhip_ctx *g_ctx = nullptr;
std::once_flag g_once;

void create_ctx()
{
	const char *dev = getenv("TRXHIP_DEVICE");
	const int rc = trxhip_create(&g_ctx, dev ? atoi(dev) : 0);
	if (rc != TRXHIP_OK) {
		fprintf(stderr, "trxarch: %s\n", trxhip_strerror(rc));
		g_ctx = nullptr;
	}
}

trxhip_ctx *ctx()
{
	std::call_once(g_once, create_ctx);
	return g_ctx;
}

/* per-thread stream and grow-only device scratch: the reference's callers run on several threads */
struct Scratch {
	hipStream_t stream = nullptr;
	void *buf[3] = { nullptr, nullptr, nullptr };
	size_t cap[3] = { 0, 0, 0 };
	void *get(int k, size_t bytes)
	{
		if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess)
			return nullptr;
		if (bytes > cap[k]) {
			if (buf[k]) (void)hipFree(buf[k]);
			buf[k] = nullptr;
			cap[k] = 0;
			if (hipMalloc(&buf[k], bytes) != hipSuccess) { buf[k] = nullptr; return nullptr; }
			cap[k] = bytes;
		}
		return buf[k];
	}
	~Scratch()
	{
		for (int k = 0; k < 3; k++) if (buf[k]) (void)hipFree(buf[k]);
		if (stream) (void)hipStreamDestroy(stream);
	}
};
thread_local Scratch tls;

bool up(void *d, const void *h, size_t n, hipStream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess; }
bool down(void *h, const void *d, size_t n, hipStream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) == hipSuccess; }

/* bounds_check(), convolve_base.c:88-105, with its messages */
int bounds_check(int x_len, int h_len, int y_len, int start, int len)
{
	if ((x_len < 1) || (h_len < 1) || (y_len < 1) || (len < 1)) {
		fprintf(stderr, "Convolve: Invalid input\n");
		return -1;
	}
	if ((start + len > x_len) || (len > y_len) || (x_len < h_len)) {
		fprintf(stderr, "Convolve: Boundary exception\n");
		fprintf(stderr, "start: %i, len: %i, x: %i, h: %i, y: %i\n", start, len, x_len, h_len, y_len);
		return -1;
	}
	return 0;
}

int convolve_on_gpu(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len, bool cplx)
{
	if (bounds_check(x_len, h_len, y_len, start, len) < 0)
		return -1;
	trxhip_ctx *c = ctx();
	if (!c || h_len > 256)
		return -1;
	Scratch &t = tls;
	/* the kernel reads samples start - (h_len-1) .. start + len - 1: upload that window only, re-based so that the
	 * first output sits at index h_len - 1 */
	const int win = len + h_len - 1;
	const float *x0 = x + 2 * ((ptrdiff_t)start - (h_len - 1));
	float *d_x = static_cast<float *>(t.get(0, (size_t)win * 8));
	float *d_h = static_cast<float *>(t.get(1, (size_t)h_len * 8));
	float *d_y = static_cast<float *>(t.get(2, (size_t)len * 8));
	if (!d_x || !d_h || !d_y)
		return -1;
	const int rc = (up(d_x, x0, (size_t)win * 8, t.stream) && up(d_h, h, (size_t)h_len * 8, t.stream))
		? (cplx ? trxhip_convolve_complex_batch(c, d_x, win, d_h, h_len, d_y, len, h_len - 1, len, 1, t.stream)
			: trxhip_convolve_real_batch(c, d_x, win, d_h, h_len, d_y, len, h_len - 1, len, 1, t.stream))
		: TRXHIP_EIO;
	if (rc != TRXHIP_OK || !down(y, d_y, (size_t)len * 8, t.stream) || hipStreamSynchronize(t.stream) != hipSuccess) {
		fprintf(stderr, "Convolve: GPU error\n");
		return -1;
	}
	return len;
}

}  // namespace

struct fft_hdl {
	float *in, *out;
	int reverse, m, howmany, istride, ostride, ooffset;
};

extern "C" {

void *convolve_h_alloc(size_t num)
{
	void *p = nullptr;
	return posix_memalign(&p, 16, (num ? num : 1) * 2 * sizeof(float)) == 0 ? p : nullptr;
}

int convolve_real(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len)
{
	return convolve_on_gpu(x, x_len, h, h_len, y, y_len, start, len, false);
}

int convolve_complex(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len)
{
	return convolve_on_gpu(x, x_len, h, h_len, y, y_len, start, len, true);
}

int base_convolve_real(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len)
{
	return convolve_on_gpu(x, x_len, h, h_len, y, y_len, start, len, false);
}

int base_convolve_complex(const float *x, int x_len, const float *h, int h_len, float *y, int y_len, int start, int len)
{
	return convolve_on_gpu(x, x_len, h, h_len, y, y_len, start, len, true);
}

void convolve_init(void) { (void)ctx(); }
void convert_init(void) { (void)ctx(); }

void convert_short_float(float *out, const short *in, int len)
{
	trxhip_ctx *c = ctx();
	Scratch &t = tls;
	if (len <= 0)
		return;
	int16_t *d_in = c ? static_cast<int16_t *>(t.get(0, (size_t)len * 2)) : nullptr;
	float *d_out = c ? static_cast<float *>(t.get(1, (size_t)len * 4)) : nullptr;
	if (!d_in || !d_out || !up(d_in, in, (size_t)len * 2, t.stream) ||
	    trxhip_convert_short_float(c, d_out, d_in, (size_t)len, t.stream) != TRXHIP_OK ||
	    !down(out, d_out, (size_t)len * 4, t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)
		fprintf(stderr, "convert_short_float: GPU error\n");
}

void convert_float_short(short *out, const float *in, float scale, int len)
{
	trxhip_ctx *c = ctx();
	Scratch &t = tls;
	if (len <= 0)
		return;
	float *d_in = c ? static_cast<float *>(t.get(0, (size_t)len * 4)) : nullptr;
	int16_t *d_out = c ? static_cast<int16_t *>(t.get(1, (size_t)len * 2)) : nullptr;
	if (!d_in || !d_out || !up(d_in, in, (size_t)len * 4, t.stream) ||
	    trxhip_convert_float_short(c, d_out, d_in, scale, (size_t)len, t.stream) != TRXHIP_OK ||
	    !down(out, d_out, (size_t)len * 2, t.stream) || hipStreamSynchronize(t.stream) != hipSuccess)
		fprintf(stderr, "convert_float_short: GPU error\n");
}

void base_convert_short_float(float *out, const short *in, int len) { convert_short_float(out, in, len); }
void base_convert_float_short(short *out, const float *in, float scale, int len) { convert_float_short(out, in, scale, len); }

struct fft_hdl *init_fft(int reverse, int m, int istride, int ostride, float *in, float *out, int ooffset)
{
	if (!ctx() || m < 1 || istride < 1 || ostride < istride || !in || !out)
		return NULL;
	struct fft_hdl *h = static_cast<struct fft_hdl *>(malloc(sizeof(struct fft_hdl)));
	if (!h)
		return NULL;
	h->in = in; h->out = out; h->reverse = reverse ? 1 : 0; h->m = m;
	h->howmany = istride;                                  /* fft.c:60: howmany = istride, idist = odist = 1 */
	h->istride = istride; h->ostride = ostride; h->ooffset = ooffset;
	return h;
}

void *fft_malloc(size_t size)
{
	void *p = nullptr;
	return posix_memalign(&p, 64, size ? size : 1) == 0 ? p : nullptr;
}

void fft_free(void *ptr) { free(ptr); }
void free_fft(struct fft_hdl *hdl) { free(hdl); }

int cxvec_fft(struct fft_hdl *h)
{
	trxhip_ctx *c = ctx();
	if (!c || !h)
		return -1;
	Scratch &t = tls;
	/* element (j, t) of the input is in[j*istride + t], t < howmany: m rows of `istride`; of the output, m rows of
	 * `ostride` starting at out + ooffset -- only the howmany touched columns of each row travel back */
	const size_t in_n = (size_t)h->m * h->istride, out_n = (size_t)h->m * h->ostride;
	float *d_in = static_cast<float *>(t.get(0, in_n * 8));
	float *d_out = static_cast<float *>(t.get(1, out_n * 8));
	if (!d_in || !d_out || !up(d_in, h->in, in_n * 8, t.stream) ||
	    trxhip_dft_batch(c, d_in, d_out, h->m, (size_t)h->howmany, (size_t)h->istride, (size_t)h->ostride, h->reverse, t.stream) != TRXHIP_OK ||
	    hipMemcpy2DAsync(h->out + 2 * (size_t)h->ooffset, (size_t)h->ostride * 8, d_out, (size_t)h->ostride * 8, (size_t)h->howmany * 8,
			     (size_t)h->m, hipMemcpyDeviceToHost, t.stream) != hipSuccess ||
	    hipStreamSynchronize(t.stream) != hipSuccess) {
		fprintf(stderr, "cxvec_fft: GPU error\n");
		return -1;
	}
	return 0;
}

}  // extern "C"
