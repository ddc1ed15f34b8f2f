"""GPU tests of libtrxarch.so: the reference's arch seam (convolve.h / convert.h / fft.h) under its own names, host
pointers in and out, every call executed on the MI355X."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_TEST = os.path.join(ROOT, "oracle", "_ref", "convolve_test_trxarch")


@pytest.fixture(scope="module")
def arch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from osmo_trx_amd import build as trx_build
    trx_build.build_all()
    L = C.CDLL(os.path.join(ROOT, "osmo_trx_amd", "lib", "libtrxarch.so"))
    for f in ("convolve_real", "convolve_complex", "base_convolve_real", "base_convolve_complex"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.convert_short_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.convert_float_short.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
    L.init_fft.restype = C.c_void_p
    L.init_fft.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.cxvec_fft.argtypes = [C.c_void_p]
    L.free_fft.argtypes = [C.c_void_p]
    L.convolve_init()
    L.convert_init()
    return L


def test_reference_convolve_test_unmodified(arch, golden_dir):
    """The reference's tests/Transceiver52M/convolve_test.c, compiled unmodified in the container and linked against
    libtrxarch.so, prints exactly the reference's convolve_test.ok when its kernels run on the GPU."""
    if not os.path.exists(CONV_TEST):
        pytest.skip("oracle/_ref/convolve_test_trxarch was not built (no /root/reference at build time)")
    out = subprocess.run([CONV_TEST], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert out == open(os.path.join(golden_dir, "convolve_test.ok")).read()


def test_convolve_with_head_room_in_front_of_x(arch):
    """The callers' convention: x points behind head room and start < h_len - 1 reads in front of x[0]
    (signalVector head room; convolve_base.c:57-85 has no lower bound check)."""
    rng = np.random.default_rng(3)
    store = rng.standard_normal((64 + 625) * 2).astype(np.float32)
    for h_len, cplx in ((16, False), (20, False), (16, True), (40, True)):
        h = rng.standard_normal(h_len * 2).astype(np.float32)
        x_ptr = store.ctypes.data + 64 * 8                      # x = store + 64 samples of head room
        y = np.zeros(625 * 2, dtype=np.float32)
        fn = arch.convolve_complex if cplx else arch.convolve_real
        assert fn(x_ptr, 625, h.ctypes.data, h_len, y.ctypes.data, 625, 0, 600) == 600
        ref = np.zeros(625 * 2, dtype=np.float32)
        # oracle on the whole store: same window, start shifted by the head room
        ofn = O.lib().orc_convolve_complex if cplx else O.lib().orc_convolve_real
        assert ofn(store.ctypes.data, 689, h.ctypes.data, h_len, ref.ctypes.data, 625, 64, 600) == 600
        assert np.array_equal(y, ref)
        # bounds failures return -1 and leave y alone (bounds_check, convolve_base.c:88-105)
        assert fn(x_ptr, 625, h.ctypes.data, h_len, y.ctypes.data, 625, 100, 600) == -1
        assert fn(x_ptr, 625, h.ctypes.data, h_len, y.ctypes.data, 10, 0, 600) == -1
    h = rng.standard_normal(40 * 2).astype(np.float32)
    y = np.zeros(625 * 2, dtype=np.float32)
    ref = np.zeros(625 * 2, dtype=np.float32)
    assert arch.base_convolve_complex(store.ctypes.data, 689, h.ctypes.data, 40, y.ctypes.data, 625, 39, 500) == 500
    assert O.lib().orc_convolve_complex(store.ctypes.data, 689, h.ctypes.data, 40, ref.ctypes.data, 625, 39, 500) == 500
    assert np.array_equal(y, ref)


def test_convert_both_ways(arch):
    rng = np.random.default_rng(4)
    s = rng.integers(-32768, 32768, 1250, dtype=np.int16)
    f = np.zeros(1250, dtype=np.float32)
    arch.convert_short_float(f.ctypes.data, s.ctypes.data, 1250)
    assert np.array_equal(f, s.astype(np.float32))              # no scaling (convert_base.c:27-31)
    x = (rng.standard_normal(1000) * 0.4).astype(np.float32)
    out = np.zeros(1000, dtype=np.int16)
    arch.convert_float_short(out.ctypes.data, x.ctypes.data, np.float32(32767.0), 1000)
    assert np.array_equal(out, np.trunc(x * np.float32(32767.0)).astype(np.int16))   # truncation (convert_base.c:20-25)


@pytest.mark.parametrize("m,reverse", [(4, 0), (4, 1), (8, 0), (3, 1)])
def test_cxvec_fft_channelizer_geometry(arch, m, reverse):
    """init_fft(reverse, m, blockLen, blockLen + hLen, in, out, hLen) as ChannelizerBase::initFFT() calls it
    (ChannelizerBase.cpp:154-155): `blockLen` transforms across the m rows, output behind hLen samples of history."""
    block, hlen = 192, 16
    rng = np.random.default_rng(m)
    xin = rng.standard_normal((m, block, 2)).astype(np.float32)
    out = np.full((m, block + hlen, 2), 5.0, dtype=np.float32)
    hdl = arch.init_fft(reverse, m, block, block + hlen, xin.ctypes.data, out.ctypes.data, hlen)
    assert hdl
    assert arch.cxvec_fft(hdl) == 0
    arch.free_fft(hdl)
    xc = xin[..., 0].astype(np.complex128) + 1j * xin[..., 1]
    ref = (np.fft.ifft(xc, axis=0) * m) if reverse else np.fft.fft(xc, axis=0)
    got = out[:, hlen:, 0] + 1j * out[:, hlen:, 1]
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    assert (out[:, :hlen] == 5.0).all()                         # history columns untouched
    if m == 4 and not reverse:
        # exact +-1 / +-j butterflies: identical to the oracle's restatement of the 4-point forward DFT
        t1, t2 = xin[0] + xin[2], xin[0] - xin[2]
        t3, t4 = xin[1] + xin[3], xin[1] - xin[3]
        assert np.array_equal(out[0, hlen:], t1 + t3) and np.array_equal(out[2, hlen:], t1 - t3)
        assert np.array_equal(out[1, hlen:, 0], t2[:, 0] + t4[:, 1]) and np.array_equal(out[1, hlen:, 1], t2[:, 1] - t4[:, 0])
