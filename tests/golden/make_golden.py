#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference tree (run in the build container only).

Produces DATA fixtures only (inputs + expected outputs), never reference source text:
  convolve_golden.npz   -- the float arrays of tests/Transceiver52M/convolve_test_golden.h
                           (the reference's own known-answer vectors) parsed to numpy
  nb_chunk_tsc7.cfile, demodbits_tsc7.s8
                        -- the reference's captured 4-SPS TSC-7 burst and its known-good bits
                           (utils/va-test/, used by burst-gen.cpp:256-290)
  ref_va_vectors.npz    -- outputs of the reference's own viterbi_detector() (oracle/_ref/libref_va.so)
  ref_arch_vectors.npz  -- outputs of the reference's own arch kernels and Resampler class,
                           compiled unmodified into oracle/_ref (oracle/Makefile), on seeded inputs:
                           convolve_real/convolve_complex (generic + SSE builds), convert_short_float,
                           Resampler(1,4) and Resampler(65,48) rotate()
Usage: python tests/golden/make_golden.py
"""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def parse_golden_header(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"static const float (\w+)\[\] = \{(.*?)\};", txt, re.S):
        vals = [float(v.rstrip("f")) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+f", m.group(2))]
        out[m.group(1)] = np.array(vals, dtype=np.float32)
    return out


def ref_lib(kind):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{kind}.so"))
    for f in (L.convolve_real, L.convolve_complex, L.base_convolve_real, L.base_convolve_complex):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.convert_short_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.convolve_h_alloc.restype = C.c_void_p
    L.convolve_h_alloc.argtypes = [C.c_size_t]
    L.ref_resampler_new.restype = C.c_void_p
    L.ref_resampler_new.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    L.ref_resampler_rotate.restype = C.c_int
    L.ref_resampler_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ref_resampler_free.argtypes = [C.c_void_p]
    L.convolve_init()
    L.convert_init()
    return L


def aligned(n_floats, align=16):
    """float32 view whose first element is 16-byte aligned (the SSE kernels use aligned loads on h)."""
    raw = np.zeros(n_floats + align, dtype=np.float32)
    off = ((16 - raw.ctypes.data % 16) % 16) // 4
    return raw[off:off + n_floats]


def make_va_vectors():
    """ref_va_vectors.npz -- the reference's own viterbi_detector() (grgsm_vitac/viterbi_detector.cc compiled
    unmodified into oracle/_ref/libref_va.so) on seeded matched-filter outputs: normal (148) and access (88) lengths,
    start states 3 / other, both stop states reachable."""
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_va.so"))
    L.ref_viterbi_detector.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p]
    rng = np.random.default_rng(0x5A17)
    out = {}
    stops = np.array([4, 12], dtype=np.uint32)
    cases = []
    for k in range(48):
        n = 148 if k % 3 else 88
        bits = rng.integers(0, 2, n) * 2 - 1
        # something Viterbi-like: +-1 symbols alternating imag / real plus ISI-ish noise, random scale
        x = np.zeros(n, dtype=np.complex64)
        x.real = np.where(np.arange(n) % 2 == 1, bits, 0) + rng.normal(0, 0.4, n)
        x.imag = np.where(np.arange(n) % 2 == 0, bits, 0) + rng.normal(0, 0.4, n)
        x *= np.float32(10 ** rng.uniform(-3, 1))
        rhh = (rng.normal(0, 0.3, 5) + 1j * rng.normal(0, 0.3, 5)).astype(np.complex64)
        rhh[0] = 1.0
        start = 3 if k % 4 else int(rng.integers(0, 16))
        y = np.zeros(n, dtype=np.float32)
        L.ref_viterbi_detector(x.ctypes.data, n, rhh.ctypes.data, start, stops.ctypes.data, 2, y.ctypes.data)
        out[f"in_{k}"], out[f"rhh_{k}"], out[f"out_{k}"] = x.view(np.float32), rhh.view(np.float32), y
        cases.append((n, start))
    out["cases"] = np.array(cases, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_va_vectors.npz"), **out)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    make_va_vectors()
    if len(sys.argv) > 1 and sys.argv[1] == "va":
        return

    g = parse_golden_header(os.path.join(REF, "tests/Transceiver52M/convolve_test_golden.h"))
    assert len(g) == 12, sorted(g)
    np.savez(os.path.join(HERE, "convolve_golden.npz"), **g)

    for f in ("nb_chunk_tsc7.cfile", "demodbits_tsc7.s8"):
        shutil.copyfile(os.path.join(REF, "utils/va-test", f), os.path.join(HERE, f))

    rng = np.random.default_rng(0x05D07A58)
    out = {}
    # convolve kernels at the hot-path shapes: taps in 16-byte aligned buffers
    x = aligned(2 * 700)
    x[:] = rng.standard_normal(1400).astype(np.float32) * 3000
    out["x"] = x.copy()
    s = rng.integers(-32768, 32768, size=1250, dtype=np.int16)
    s[:4] = [-32768, 32767, 0, -1]
    out["cvt_in"] = s
    for kind in ("generic", "sse"):
        L = ref_lib(kind)
        for h_len in (4, 8, 12, 16, 20, 24, 40, 64, 5, 1):
            h = aligned(2 * h_len)
            h[:] = np.random.default_rng(h_len).standard_normal(2 * h_len).astype(np.float32)
            out[f"h_{h_len}"] = h.copy()
            start, ln = h_len - 1, 700 - (h_len - 1)
            for name, fn in (("real", L.convolve_real), ("complex", L.convolve_complex)):
                y = aligned(2 * 700)
                y[:] = 0
                rc = fn(x.ctypes.data, 700, h.ctypes.data, h_len, y.ctypes.data, 700, start, ln)
                assert rc == ln
                out[f"y_{kind}_{name}_{h_len}"] = y[: 2 * ln].copy()
        # int16 -> fp32
        f = aligned(1250)
        L.convert_short_float(f.ctypes.data, s.ctypes.data, 1250)
        out[f"cvt_{kind}"] = f.copy()
        # Resampler(1,4): the /4 decimator of sigProcLib (624 -> 156), 16 samples of zero history
        for (p, q, n_in, n_out, tag) in ((1, 4, 624, 156, "dec4"), (65, 48, 192, 260, "rs6548")):
            h_ = L.ref_resampler_new(p, q, 16, 1.0)
            buf = aligned(2 * (16 + n_in))
            src = np.random.default_rng(p * 1000 + q).standard_normal(2 * (16 + n_in)).astype(np.float32) * 1000
            if tag == "dec4":
                src[:32] = 0
            buf[:] = src
            y = aligned(2 * n_out)
            rc = L.ref_resampler_rotate(h_, buf[32:].ctypes.data, n_in, y.ctypes.data, n_out)
            assert rc == n_out
            out[f"{tag}_in"] = buf.copy()
            out[f"{tag}_{kind}"] = y.copy()
            # impulse responses expose the partition taps
            if tag == "dec4":
                imp = aligned(2 * (16 + 624))
                imp[:] = 0
                imp[2 * (16 + 300)] = 1.0
                y2 = aligned(2 * 156)
                L.ref_resampler_rotate(h_, imp[32:].ctypes.data, 624, y2.ctypes.data, 156)
                out[f"dec4_impulse_{kind}"] = y2.copy()
            L.ref_resampler_free(h_)
    np.savez_compressed(os.path.join(HERE, "ref_arch_vectors.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
