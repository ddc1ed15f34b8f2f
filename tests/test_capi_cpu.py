"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/trxhip.h declares, its host-generated table blob is bit-identical to the oracle's tables, and
it refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from osmo_trx_amd import trxhip
from osmo_trx_amd import build as trx_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    trx_build.build_lib()
    return trxhip.load_library()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "trxhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(trxhip_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/trxhip.h but not exported"
    assert sorted(trxhip.SYMBOLS) == names
    assert lib.trxhip_abi_version() == 5


# device blob layout (osmo_trx_amd/csrc/trx_tables.h)
SEQ = np.dtype([("taps", "<c8", 64), ("gain", "<c8"), ("gain_inv", "<c8"), ("ci_den", "<f4"), ("toa", "<f4"),
                ("n", "<i4"), ("ci_den_inv", "<f4")])
BLOB = np.dtype([("magic", "<u4"), ("version", "<u4"), ("dec_taps", "<f4", 16), ("delay_filt", "<f4", (64, 20)),
                 ("rrot1", "<c8", 160), ("c0_inv", "<f4", 8), ("seq", SEQ, 21), ("sincv", "<f4", 4096),
                 ("chan_taps", "<f4", (4, 16)), ("rs6548_taps", "<f4", (65, 16)),
                 ("comp_filt", "<f4", (65, 36)), ("edge_derot", "<c8", 16), ("edge_ideal", "<c8", 9),
                 ("edge_rot2", "<c8", 2), ("edge_step", "<f4"), ("edge_pad", "<f4"),
                 ("unit_neg", "<u8", 21), ("unit_ok", "<u4"), ("unit_pad", "<u4"),
                 ("edge_lo", "<f4", (65, 15, 36)), ("fused_u0", "<u4"), ("fused_nt", "<u4"), ("edge8", "<f4", (65, 8, 24)),
                 ("edge_hi", "<f4", (65, 15, 36))])


def test_tables_bit_identical_to_oracle(lib):
    blob = trxhip.generate_tables_host()
    assert len(blob) == BLOB.itemsize == lib.trxhip_tables_size()
    t = np.frombuffer(blob, dtype=BLOB)[0]
    o = O.tables()
    assert t["magic"] == 0x54585254
    assert np.array_equal(t["dec_taps"], o["dec_taps"])
    assert np.array_equal(t["delay_filt"], o["delay_filt"])
    # taps 0, 17, 18, 19 of every fractional-delay filter are exactly zero (sinc LUT zero beyond 8 pi): the exact demodulator
    # skips them (trx_kernel4.hip, K0 / K1)
    assert not o["delay_filt"][:, [0, 17, 18, 19]].any()
    assert np.array_equal(t["rrot1"][:157].view(np.float32), o["rrot1"].view(np.float32))
    assert np.array_equal(t["c0_inv"][:5], o["c0_inv"])

    def same(seq, ref):
        assert seq["n"] == ref["n"]
        assert np.array_equal(seq["taps"][: ref["n"]].view(np.float32), ref["seq"].view(np.float32))
        assert np.complex64(seq["gain"]) == np.complex64(ref["gain"])
        assert np.float32(seq["toa"]) == np.float32(ref["toa"])

    for i in range(8):
        same(t["seq"][i], o["midamble"][i])
        same(t["seq"][12 + i], o["edge_midamble"][i])
    for i in range(3):
        same(t["seq"][8 + i], o["rach"][i])
    same(t["seq"][11], o["dummy"])
    same(t["seq"][20], o["sch"])
    # derived tables: sincv[q ^ ((q>>4)&31)] = sinc LUT of M_PI_F*q/512 (sigProcLib.cpp:990-998)
    st = o["sinc_table"]
    pi_f = np.float32(np.pi)
    for q in list(range(0, 4096, 37)) + [0, 1, 4, 511, 512, 2048, 4095]:
        x = np.float32(pi_f * np.float32(q / 512.0))
        v = np.float64(abs(x)) / (8 * np.pi) * 1024
        ref = 0.0 if np.float64(abs(x)) >= 8 * np.pi else st[int(np.floor(np.float32(v)))]
        assert t["sincv"][q ^ ((q >> 4) & 31)] == np.float32(ref), q
    # composite (delay o decimate) filters of the fused demodulator: 35 taps, sum = 1, row 64 = shifted decimator
    for f in (0, 1, 32, 63):
        ref = np.convolve(o["dec_taps"].astype(np.float64), o["delay_filt"][f].astype(np.float64))
        np.testing.assert_allclose(t["comp_filt"][f][:35], ref, rtol=1e-7, atol=1e-12)   # products in double, one float rounding
        assert t["comp_filt"][f][35] == 0
    assert np.array_equal(t["comp_filt"][64][9:25], o["dec_taps"]) and not t["comp_filt"][64][:9].any()
    # the fused demodulator runs taps 6 .. 29 only (trx_tables.h TRX_FUSED_U0 / TRX_FUSED_NT): what it leaves out is below 8e-7
    # of a filter whose taps sum to 1, in every row (taps 8 .. 31, rounds 2-4: 1.1e-6).  (Not so for the truncated rows of the
    # low-edge table, which keep all taps.)
    assert np.abs(t["comp_filt"][:, :8]).sum(axis=1).max() < 1.1e-6 and not t["comp_filt"][:, 32:].any()
    assert (np.abs(t["comp_filt"][:, :6]).sum(axis=1) + np.abs(t["comp_filt"][:, 30:]).sum(axis=1)).max() < 8e-7
    # truncated composites of the low-side partial outputs: decimator taps t >= t0 only
    for f in (0, 17, 63):
        for t0 in (1, 3, 7, 11, 15):
            g = o["dec_taps"].astype(np.float64).copy()
            g[:t0] = 0.0
            ref = np.convolve(g, o["delay_filt"][f].astype(np.float64))
            np.testing.assert_allclose(t["edge_lo"][f][t0 - 1][:35], ref, rtol=1e-7, atol=1e-12)
            assert t["edge_lo"][f][t0 - 1][35] == 0
    assert np.array_equal(t["edge_lo"][64][6][9 + 7:25], o["dec_taps"][7:]) and not t["edge_lo"][64][6][:16].any()
    # ... and of the high-side ones (bursts shifted left by a large TOA: access bursts): decimator taps t <= tm only
    for f in (0, 17, 63):
        for tm in (0, 2, 7, 11, 14):
            g = o["dec_taps"].astype(np.float64).copy()
            g[tm + 1:] = 0.0
            ref = np.convolve(g, o["delay_filt"][f].astype(np.float64))
            np.testing.assert_allclose(t["edge_hi"][f][tm][:35], ref, rtol=1e-7, atol=1e-12)
            assert t["edge_hi"][f][tm][35] == 0
    assert np.array_equal(t["edge_hi"][64][6][9:16], o["dec_taps"][:7]) and not t["edge_hi"][64][6][16:].any()
    # low + high truncations at the same cut add up to the full composite
    np.testing.assert_allclose(t["edge_lo"][:, 7, :] + t["edge_hi"][:, 7, :], t["comp_filt"], rtol=0, atol=2e-7)
    # the usual-geometry repack the main filter loop consumes (trx_tables.h, edge8): rows of outputs 0..3 from tap 6 on,
    # their taps below 6 as separate rows; nothing is lost by the 32-tap window (tap 0 and taps >= 32 are exactly zero)
    assert not t["edge_lo"][:, :, 32:].any() and not t["edge_lo"][:, :, 0].any()
    for i in range(4):
        assert np.array_equal(t["edge8"][:, i, :], t["edge_lo"][:, 14 - 4 * i, 6:30])                 # taps u = 6 .. 29
        assert np.array_equal(t["edge8"][:, 4 + i, 2:8], t["edge_lo"][:, 14 - 4 * i, :6])              # its taps u < 6, window 8 samples early
        assert not t["edge8"][:, 4 + i, :2].any() and not t["edge8"][:, 4 + i, 8:].any()
    # the straight-line decimator mirrors taps 0..7 (trx_kernel4.hip, decimate16_sym): bitwise symmetric filter
    assert np.array_equal(t["dec_taps"].view(np.uint32), t["dec_taps"][::-1].view(np.uint32))
    # resampler / channelizer partitions against the oracle's restatements
    L = O.lib()
    r = L.orc_resampler_new(65, 48, 16, 1.0)
    for path in (0, 1, 17, 64):
        ref = np.ctypeslib.as_array(L.orc_resampler_partition(r, path), shape=(16,))
        assert np.array_equal(t["rs6548_taps"][path], ref)
    L.orc_resampler_free(r)
    c = L.orc_channelizer_new(4, 192, 16)
    for path in range(4):
        ref = np.ctypeslib.as_array(L.orc_channelizer_subfilter(c, path), shape=(16,))
        assert np.array_equal(t["chan_taps"][path], ref)
    L.orc_channelizer_free(c)


def test_checksum_is_stable(lib):
    a = trxhip.generate_tables_host()
    b = trxhip.generate_tables_host()
    assert a == b and trxhip.tables_checksum(a) == trxhip.tables_checksum(b)
    assert trxhip.tables_checksum(a[:-1] + bytes([a[-1] ^ 1])) != trxhip.tables_checksum(a)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.trxhip_create(C.byref(h), 0) == -19          # TRXHIP_ENODEV
    assert lib.trxhip_device_count() == 0
    with pytest.raises(trxhip.TrxHipError):
        trxhip.TrxHip(0)


def test_product_never_imports_oracle():
    """The shipped package must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "osmo_trx_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "trx_oracle" not in txt and "liboracle" not in txt, f


def test_trxarch_exports_the_reference_seam():
    """libtrxarch.so loads without a GPU and exports every symbol include/trxarch.h declares -- the names of
    arch/common/convolve.h:4-26, convert.h:4-13, fft.h:6-11."""
    trx_build.build_all()
    txt = open(os.path.join(ROOT, "include", "trxarch.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\(", txt)))
    assert names == sorted(["convolve_h_alloc", "convolve_real", "convolve_complex", "base_convolve_real",
                            "base_convolve_complex", "convolve_init", "convert_float_short", "convert_short_float",
                            "base_convert_float_short", "base_convert_short_float", "convert_init", "init_fft",
                            "fft_malloc", "fft_free", "free_fft", "cxvec_fft"])
    L = C.CDLL(os.path.join(ROOT, "osmo_trx_amd", "lib", "libtrxarch.so"))
    for n in names:
        assert hasattr(L, n), n
    import torch
    if not torch.cuda.is_available():
        # no CPU fallback: the reference's error value, not a computed result
        x = np.zeros(64, dtype=np.float32)
        y = np.full(16, 7.0, dtype=np.float32)
        L.convolve_real.restype = C.c_int
        assert L.convolve_real(x.ctypes.data, 32, x.ctypes.data, 4, y.ctypes.data, 8, 3, 8) == -1
        assert (y == 7.0).all()


def test_unit_tap_structure_of_gmsk_sequences(lib):
    """The premise of corr_unit() (csrc/trx_device.h): every tap of the GMSK correlation sequences is (+-1, e) for even k,
    (e, +-1) for odd k, |e| <= 5e-14 -- checked here against the ORACLE's tables -- and the sign patterns compiled into
    the kernels are the ones the tables have."""
    t = np.frombuffer(trxhip.generate_tables_host(), dtype=BLOB)[0]
    o = O.tables()
    hdr = open(os.path.join(ROOT, "osmo_trx_amd", "csrc", "trx_device.h")).read()
    compiled = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define TRX_UNIT_NEG_(\w+)\s+0x([0-9a-f]+)ull", hdr)}
    seqs = [("TSC%d" % i, i, o["midamble"][i]) for i in range(8)] + [("RACH%d" % i, 8 + i, o["rach"][i]) for i in range(3)] + \
        [("DUMMY", 11, o["dummy"]), ("SCH", 20, o["sch"])]           # (SCH: 64 taps, trx_sch.hip; residue up to 7e-14 at its last taps)
    for name, s, ref in seqs:
        taps = ref["seq"]
        neg = 0
        for k, h in enumerate(taps):
            one, eps = (h.imag, h.real) if k & 1 else (h.real, h.imag)
            assert abs(one) == 1.0 and abs(eps) <= (1e-13 if name == "SCH" else 5e-14), (name, k, h)
            if one < 0:
                neg |= 1 << k
        assert (int(t["unit_ok"]) >> s) & 1
        assert int(t["unit_neg"][s]) == neg == compiled[name], name
    # 2^17 * max|e| stays below a quarter ulp (2^-26): the guard of the exactness argument
    emax = max(max(abs(h.real) if k & 1 else abs(h.imag) for k, h in enumerate(ref["seq"])) for _, _, ref in seqs)
    assert (1 << 17) * emax < 2.0 ** -26
    # 8-PSK midambles do not have the structure and must not be flagged
    for i in range(8):
        assert not (int(t["unit_ok"]) >> (12 + i)) & 1
