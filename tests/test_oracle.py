"""CPU tests: pin the oracle (oracle/trx_oracle.c) against everything the reference holds for the path.

  * the reference's own known-answer vectors (tests/Transceiver52M/convolve_test_golden.h -> convolve_golden.npz)
    with the test's LCG inputs restated (convolve_test.c:15-45) and its tolerance (1e-5 abs OR rel, :76-96)
  * outputs of the reference's arch kernels + Resampler compiled unmodified (oracle/_ref -> ref_arch_vectors.npz):
    bit-exact vs the generic-C build, <=1e-5 vs the SSE build
  * the reference's captured burst + known-good bits (utils/va-test)
  * the anchor values the survey dumped from the compiled reference
"""
import os

import numpy as np
import pytest

import oracle_lib as O


# ---- reference test's input generator, convolve_test.c:15-45 ----
def lcg_floats(n, state=0):
    out = np.zeros(n, dtype=np.float32)
    u32 = np.zeros(1, dtype=np.uint32)
    for i in range(n):
        state = (1103515245 * state + 12345) & 0x7FFFFFFF
        u = state
        e = 112 + ((u ^ (u >> 8)) & 15)
        r = (u & 0x007FFFFF) | ((u & 0x00800000) << 8) | ((e & 0xFF) << 23)
        u32[0] = r
        out[i] = u32.view(np.float32)[0]
    return out, state


def ref_close(a, b, delta=1e-5, eps=1e-5):
    """compare_floats(), convolve_test.c:76-96"""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ok = (np.abs(a - b) < delta) | (np.abs(1.0 - a / b) < eps)
    return bool(ok.all())


@pytest.mark.parametrize("h_len", [4, 8, 12, 16, 20, 24])
@pytest.mark.parametrize("kind", ["real", "complex"])
def test_convolve_reference_known_answer(golden_dir, h_len, kind):
    g = np.load(os.path.join(golden_dir, "convolve_golden.npz"))
    x, st = lcg_floats(200)
    h, _ = lcg_floats(50, st)
    start, ln = h_len - 1, 100 - (h_len - 1)
    y = O.convolve(x.view(np.complex64), h.view(np.complex64)[:h_len], start, ln, kind == "complex")
    ref = g[f"y_ref_{kind}_base_{h_len}"]
    assert len(ref) == 2 * ln
    assert ref_close(ref, y.view(np.float32))


@pytest.mark.parametrize("h_len", [1, 4, 5, 8, 12, 16, 20, 24, 40, 64])
@pytest.mark.parametrize("kind", ["real", "complex"])
def test_convolve_vs_compiled_reference(golden_dir, h_len, kind):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    x = v["x"].view(np.complex64)
    h = v[f"h_{h_len}"].view(np.complex64)
    start, ln = h_len - 1, 700 - (h_len - 1)
    y = O.convolve(x, h, start, ln, kind == "complex").view(np.float32)
    # generic-C build of the reference: same operand order -> bit-exact
    assert np.array_equal(y, v[f"y_generic_{kind}_{h_len}"])
    # SSE build: different summation order, reference's own tolerance scaled to the data
    ref = v[f"y_sse_{kind}_{h_len}"]
    assert np.max(np.abs(y - ref)) <= 1e-5 * np.max(np.abs(ref))


def test_convert_short_float(golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    s = v["cvt_in"]
    out = np.zeros(len(s), dtype=np.float32)
    O.lib().orc_convert_short_float(out.ctypes.data, s.ctypes.data, len(s))
    assert np.array_equal(out, v["cvt_generic"])
    assert np.array_equal(out, v["cvt_sse"])


def test_decimator_matches_reference_resampler(golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    L = O.lib()
    r = L.orc_resampler_new(1, 4, 16, 1.0)
    buf = np.ascontiguousarray(v["dec4_in"])
    y = np.zeros(2 * 156, dtype=np.float32)
    L.orc_resampler_rotate(r, buf[32:].ctypes.data, 624, y.ctypes.data, 156)
    assert np.array_equal(y, v["dec4_generic"])
    assert np.max(np.abs(y - v["dec4_sse"])) <= 1e-5 * np.max(np.abs(y))
    # impulse at input 300 -> out[i] = g[315-4i]: exposes the reversed partition taps
    imp = v["dec4_impulse_generic"].view(np.complex64)
    taps = O.tables()["dec_taps"]
    for i in range(75, 79):
        assert imp[i].real == taps[315 - 4 * i]
    L.orc_resampler_free(r)


def test_resampler_65_48_matches_reference(golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_arch_vectors.npz"))
    L = O.lib()
    r = L.orc_resampler_new(65, 48, 16, 1.0)
    buf = np.ascontiguousarray(v["rs6548_in"])
    y = np.zeros(2 * 260, dtype=np.float32)
    L.orc_resampler_rotate(r, buf[32:].ctypes.data, 192, y.ctypes.data, 260)
    assert np.array_equal(y, v["rs6548_generic"])
    assert np.max(np.abs(y - v["rs6548_sse"])) <= 1e-5 * np.max(np.abs(y))
    L.orc_resampler_free(r)


# ---- values dumped from the compiled reference (SSE build) ----
def test_table_anchors():
    t = O.tables()
    gains = [(14.958575, 0.003145), (14.871485, -0.014816), (14.876348, -0.075098), (14.888629, -0.075098),
             (14.943010, 0.055678), (14.956352, -0.048295), (15.077748, 0.012211), (15.067246, 0.084153)]
    toas = [-1, -1, -1, -1, 1, 1, 1, -1]
    for i in range(8):
        m = t["midamble"][i]
        assert m["n"] == 16
        assert abs(m["gain"].real - gains[i][0]) < 3e-6 and abs(m["gain"].imag - gains[i][1]) < 3e-6
        assert m["toa"] == toas[i] / 512.0
    rach = [((35.715374, 0.155127), -0.005859), ((35.396175, -0.272141), -0.009766), ((35.331802, 0.108754), -0.005859)]
    for i in range(3):
        m = t["rach"][i]
        assert m["n"] == 40
        assert abs(m["gain"].real - rach[i][0][0]) < 1e-5 and abs(m["gain"].imag - rach[i][0][1]) < 1e-5
        assert abs(m["toa"] - rach[i][1]) < 1e-6
    assert t["sch"]["n"] == 64
    assert abs(t["sch"]["gain"].real - 57.088936) < 3e-5 and abs(t["sch"]["gain"].imag + 0.002823) < 1e-5
    assert abs(t["sch"]["toa"] + 0.009766) < 1e-6
    assert abs(t["dummy"]["gain"].real - 13.659099) < 3e-6 and t["dummy"]["toa"] == 1 / 512.0
    for i in range(8):
        g = t["edge_midamble"][i]["gain"]
        assert abs(g.real + 16.646780) < 2e-6 and abs(g.imag - 16.525934) < 2e-6
    s = t["midamble"][0]["seq"]
    assert s[0] == complex(-1, 0) and abs(s[1] - complex(6.12e-17, -1)) < 1e-18 and s[2].real == 1.0
    np.testing.assert_allclose(t["pulse1_c0"], [0.0052078, 0.7070876, 0.7070876, 0.0052078], atol=5e-8)
    dec = [-0.0000011, -0.0001773, -0.0015542, -0.0030354, 0.0101529, 0.0666594, 0.1694168, 0.2585389]
    np.testing.assert_allclose(t["dec_taps"][:8], dec, atol=5e-8)
    np.testing.assert_array_equal(t["dec_taps"][:8], t["dec_taps"][::-1][:8])
    assert abs(float(t["dec_taps"].astype(np.float64).sum()) - 1.0) < 1e-6
    df0 = t["delay_filt"][0]
    assert df0[9] == 1.0 and np.all(np.abs(np.delete(df0, 9)) < 1e-15)
    df32 = [0, -0.000091, 0.000668, -0.002866, 0.009240, -0.024902, 0.060830, -0.151246, 0.588012, 0.667994,
            -0.222665, 0.117602, -0.064820, 0.033794, -0.015847, 0.006397, -0.002102, 0, 0, 0]
    np.testing.assert_allclose(t["delay_filt"][32], df32, atol=6e-7)
    st = t["sinc_table"]
    assert st[0] == 1.0 and abs(st[1] - 0.9998996) < 5e-8 and abs(st[128]) < 1e-15 and abs(st[1023] + 0.0009774) < 5e-8
    rr = t["rrot1"]
    assert rr[0] == 1 and abs(rr[1] - complex(6.12e-17, -1)) < 1e-18 and rr[2].real == -1 and rr[3].imag == 1


def test_captured_burst_known_answer(golden_dir):
    """burst-gen.cpp:256-290: detectAnyBurst(sv, 7, 4.0, 4, TSC, 40) + demodAnyBurst on the 1500-sample capture."""
    x = np.fromfile(os.path.join(golden_dir, "nb_chunk_tsc7.cfile"), dtype=np.complex64)
    bits = np.fromfile(os.path.join(golden_dir, "demodbits_tsc7.s8"), dtype=np.int8)
    assert len(x) == 1500 and len(bits) == 148
    rc, e = O.detect_any_burst(x, 7, 4.0, 4, O.TSC, 40)
    assert rc == O.TSC
    # SURVEY.md Appendix A (compiled reference, SSE build)
    assert e.toa == np.float32(12.535156)
    assert abs(e.amp[0] + 0.00112989) < 1e-8 and abs(e.amp[1] - 0.00166411) < 1e-8
    assert abs(e.ci - 6.460016) < 1e-5
    soft = O.demod_any_burst(x, rc, 4, e)
    assert len(soft) == 156
    assert np.array_equal(soft[:148] > 0, bits > 0)          # 0/148 bit errors, SoftVector::bit() = (v > 0)


def test_detect_rejects_noise_and_flags_clipping():
    rng = np.random.default_rng(1)
    noise = (rng.standard_normal(625) + 1j * rng.standard_normal(625)).astype(np.complex64) * 100
    rc, e = O.detect_any_burst(noise, 0, 4.0, 4, O.TSC, 3)
    assert rc == 0 and e.toa == 0 and e.amp[0] == 0
    noise[100] = 31000
    rc, e = O.detect_any_burst(noise, 0, 4.0, 4, O.TSC, 3)
    assert rc == -O.SIGERR_CLIP
    rc, _ = O.detect_any_burst(noise, 9, 4.0, 4, O.TSC, 3)
    assert rc == -O.SIGERR_UNSUPPORTED


def test_vector_slicer_and_trxd_packing():
    src = np.array([-2.0, -1.0, -0.5, 0.0, 0.25, 1.0, 3.0], dtype=np.float32)
    dst = np.zeros_like(src)
    O.lib().orc_vector_slicer(dst.ctypes.data, src.ctypes.data, len(src))
    np.testing.assert_array_equal(dst, [0, 0, 0.25, 0.5, 0.625, 1, 1])
    assert O.lib().orc_trxd_toa256(1.5) == 384 and O.lib().orc_trxd_toa256(-0.25) == -63
    assert O.lib().orc_trxd_ci_cb(9.589) == 96
    u8 = np.zeros(len(dst), dtype=np.uint8)
    O.lib().orc_trxd_soft_u8(u8.ctypes.data, dst.ctypes.data, len(dst))
    np.testing.assert_array_equal(u8, [0, 0, 64, 128, 159, 255, 255])


# ---- detectSCHBurst (sigProcLib.cpp:1805-1861) -----------------------------------------------------------------
def test_sch_detect_full_and_demod():
    """SCH_DETECT_FULL finds a synchronisation burst at the delay it was given, and demodAnyBurst(SCH) on the
    detected parameters returns its 148 bits."""
    import sch_util
    rng = np.random.default_rng(7)
    toas = []
    for d in (0, 3, 10, 17):
        y, bits = sch_util.sch_burst(rng, 700, d)
        rc, e = O.detect_sch_burst(y, 4.0, 4, O.SCH_DETECT_FULL)
        assert rc == 1
        toas.append(e.toa)
        soft = O.demod_any_burst(y[:625], 4, 4, e)            # CorrType SCH = 4 -> demodGmskBurst
        # the correlation sequences are built from -(2b-1) (sigProcLib.cpp:1257-1260), so amp
        # carries the sign and a demodulated positive value is a transmitted 0
        assert np.array_equal((soft[:148] < 0).astype(np.uint8), bits)
    # the TOA follows the delay: 4 samples per symbol
    d = np.diff(np.array(toas))
    assert np.allclose(d, np.array([3, 7, 7]) / 4.0, atol=0.02)


def test_sch_detect_states():
    import sch_util
    rng = np.random.default_rng(8)
    y, _ = sch_util.sch_burst(rng, 700, 5)
    # noise only: nothing found, toa and amp zeroed (:1846-1850)
    z, _ = sch_util.sch_burst(rng, 700, 5, present=False)
    rc, e = O.detect_sch_burst(z, 4.0, 4, O.SCH_DETECT_FULL)
    assert rc == 0 and e.toa == 0.0 and e.amp[0] == 0.0 and e.amp[1] == 0.0
    # NARROW decimates only 32 samples and then correlates 101 symbols into that 8-sample vector (:1820-1823,
    # :1840-1842): it cannot detect anything -- restated as is
    rc, e = O.detect_sch_burst(y, 4.0, 4, O.SCH_DETECT_NARROW)
    assert rc == 0
    # sps other than 1 / 4 and short buffers are errors
    assert O.detect_sch_burst(y, 4.0, 2, O.SCH_DETECT_FULL)[0] == -1
    assert O.detect_sch_burst(y[:600], 4.0, 4, O.SCH_DETECT_FULL)[0] == -1
    # BUFFER: 12 frames, the burst anywhere inside; toa is relative to the burst start like FULL's
    buf, _ = sch_util.sch_burst(rng, 60000, 4 * 5000 + 5, noise=100.0)
    rc, e = O.detect_sch_burst(buf, 4.0, 4, O.SCH_DETECT_BUFFER)
    rc_f, e_f = O.detect_sch_burst(buf[4 * 5000:4 * 5000 + 700], 4.0, 4, O.SCH_DETECT_FULL)
    assert rc == 1 and rc_f == 1
    # BUFFER subtracts 3+39+64 = 106 where FULL subtracts head = 105 (:1853-1858): one symbol apart by construction
    assert abs((e.toa - 5000) - (e_f.toa - 1.0)) < 0.02


# ---- Viterbi alternative (cfg->use_va): Transceiver.cpp:620-645 over grgsm_vitac/ --------------------------------
def test_va_viterbi_matches_compiled_reference(golden_dir):
    """orc_va_viterbi() (table-driven restatement) == the reference's viterbi_detector() compiled unmodified
    (oracle/_ref/libref_va.so, vectors in ref_va_vectors.npz), bit for bit, incl. the soft magnitudes."""
    v = np.load(os.path.join(golden_dir, "ref_va_vectors.npz"))
    for k, (n, start) in enumerate(v["cases"]):
        out = O.va_viterbi(v[f"in_{k}"].view(np.complex64), v[f"rhh_{k}"].view(np.complex64), int(start))
        assert len(out) == n
        assert np.array_equal(out.view(np.uint32), v[f"out_{k}"].view(np.uint32)), k


def _va_burst(rng, tsc, delay_samples, snr_db=25.0, amp=8000.0):
    """625-sample 4-SPS normal burst as the VA path sees it.  Its buffer starts 20 samples (5 symbols) before the one
    the detector looks at (Transceiver.cpp:760-762, osmo-trx.cpp:97), so a burst placed `delay_samples` in shows up
    there with TOA = delay/4 - 0.4 symbols; get_norm_chan_imp_resp() tracks starts of 0 .. 20 samples."""
    from osmo_trx_amd.synth import TSC_BITS
    tsc_bits = np.array([int(c) for c in TSC_BITS[tsc]], dtype=np.uint8)
    bits = rng.integers(0, 2, 148, dtype=np.uint8)
    bits[:3] = 0
    bits[-3:] = 0
    bits[61:87] = tsc_bits
    x = O.modulate_burst(bits, 8, 4)
    y = np.zeros(625, dtype=np.complex64)
    off = delay_samples
    m = min(len(x), 625 - off)
    y[off:off + m] = x[:m] * np.complex64(amp * np.exp(1j * rng.uniform(0, 2 * np.pi)))
    sigma = amp * 10 ** (-snr_db / 20) / np.sqrt(2)
    y += ((rng.normal(size=625) + 1j * rng.normal(size=625)) * sigma).astype(np.complex64)
    return y, bits


def test_va_demodulates_normal_bursts():
    """End to end: the MLSE receiver returns the transmitted bits of a clean Laurent-GMSK burst for every TSC and a
    range of delays; output format of demodAnyBurst_va (+-127, 8 trailing zeros)."""
    rng = np.random.default_rng(31)
    errs = 0
    for k in range(32):
        tsc = k % 8
        d = int(rng.integers(0, 19))
        y, bits = _va_burst(rng, tsc, d)
        start, soft = O.demod_any_burst_va(y, O.TSC, tsc, 3)
        assert abs(start - (d + 1)) <= 6                           # 5-symbol energy window: coarse by design
        assert set(np.unique(soft[:148])) <= {-127.0, 127.0} and not soft[148:].any()
        hard = (soft[:148] > 0).astype(np.uint8)
        errs += int((hard != bits).sum())
    assert errs <= 8, errs                                     # 32 x 148 bits at 25 dB SNR


def test_va_access_burst_and_quirks():
    """RACH branch: get_access_imp_resp() + 88-bit detect_burst_ab(); rach_max_toa lands in the Viterbi start state
    (Transceiver.cpp:633): values >= 16 select no start state here (the reference stores out of bounds).  Bits 88..147
    of the output are zero here (uninitialised in the reference)."""
    rng = np.random.default_rng(33)
    ab = np.array([int(c) for c in "01001011011111111001100110101010001111000"], dtype=np.uint8)
    errs = 0
    for k in range(16):
        bits = np.concatenate([np.array([0, 0, 1, 1, 1, 0, 1, 0], np.uint8), ab, rng.integers(0, 2, 36, dtype=np.uint8),
                               np.zeros(3, np.uint8)])
        x = O.modulate_burst(bits, 8, 4)
        off = int(rng.integers(2, 20))                             # tracked starts: 0 .. 20 samples = offsets 4 .. 24 here
        y = np.zeros(625, dtype=np.complex64)
        m = min(len(x), 625 - off)
        y[off:off + m] = x[:m] * np.complex64(5000.0 * np.exp(1j * rng.uniform(0, 6.28)))
        y += ((rng.normal(size=625) + 1j * rng.normal(size=625)) * 100).astype(np.complex64)
        for max_toa in (3, 63):
            start, soft = O.demod_any_burst_va(y, O.RACH, 0, max_toa)
            assert set(np.unique(soft[:88])) <= {-127.0, 127.0} and not soft[88:].any()
            if max_toa == 3:
                errs += int(((soft[:88] > 0).astype(np.uint8) != bits).sum())
    assert errs <= 8, errs
    assert O.demod_any_burst_va(y, O.TSC, 9, 3)[0] == -1          # tsc > 7


# ---- the oracle's call graph over the reference's own compiled arch kernels (oracle/_ref, orc_set_arch) ----
def _workloads():
    from osmo_trx_amd import synth
    w = []
    iq, p, _ = synth.make_normal_bursts(192, "cpu", 4, seed=501)
    w.append(("normal, max_toa 3", iq.numpy(), p, 4))
    iq, p, _ = synth.make_normal_bursts(96, "cpu", 4, seed=502, max_toa=40, delay_sym=(-2.0, 38.0))
    w.append(("normal, max_toa 40", iq.numpy(), p, 4))
    iq, p, _ = synth.make_access_bursts(96, "cpu", seed=503)
    w.append(("access", iq.numpy(), p, 4))
    iq, p, _ = synth.make_access_bursts(64, "cpu", seed=504, ext=True)
    w.append(("extended access", iq.numpy(), p, 4))
    iq, p, _ = synth.make_edge_bursts(64, "cpu", seed=505)
    w.append(("EDGE", iq.numpy(), p, 4))
    iq, p, _ = synth.make_normal_bursts(64, "cpu", 1, seed=506, burst_len=156)
    w.append(("1 SPS", iq.numpy(), p, 1))
    return w


def _tables_bytes():
    t = O.lib().orc_get_tables().contents
    return bytes(t)


@pytest.mark.skipif(not O.ref_arch_available("generic"), reason="oracle/_ref not built (no /root/reference at build time)")
def test_call_graph_over_the_references_generic_kernels_is_bit_identical():
    """Every FIR of the detect / demod call graph and of the table generator executed by the reference's OWN convolve_real /
    convolve_complex (arch/common/convolve_base.c through arch/x86/convolve.c, compiled unmodified: oracle/_ref/
    libref_generic.so) instead of the oracle's restated loops: tables byte-identical, every result record and every soft bit
    identical, for normal / access / extended access / EDGE / 1-SPS bursts and the captured burst.  This pins the restated
    kernels, the zero padding of convolve() (sigProcLib.cpp:309-358) and the arguments of every call site against the
    reference's objects inside the full call graph -- not only stage by stage."""
    loads = _workloads()
    cap = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nb_chunk_tsc7.cfile"), dtype=np.complex64)
    try:
        O.use_ref_arch(None)
        base_t = _tables_bytes()
        base = [O.pull_batch(iq, sps, p, soft_stride=444 if name == "EDGE" else 156, slice_bits=False) for name, iq, p, sps in loads]
        rc0, e0 = O.detect_any_burst(cap, 7, 4.0, 4, O.TSC, 40)
        s0 = O.demod_any_burst(cap, rc0, 4, e0)
        O.use_ref_arch("generic")
        assert _tables_bytes() == base_t
        for (name, iq, p, sps), (r0, sf0) in zip(loads, base):
            r1, sf1 = O.pull_batch(iq, sps, p, soft_stride=444 if name == "EDGE" else 156, slice_bits=False)
            assert r1.tobytes() == r0.tobytes(), name
            assert np.array_equal(sf1.view(np.uint32), sf0.view(np.uint32)), name
            assert (r0["rc"] > 0).sum() > 0.7 * len(p), name
        rc1, e1 = O.detect_any_burst(cap, 7, 4.0, 4, O.TSC, 40)
        assert rc1 == rc0 == O.TSC and bytes(e1) == bytes(e0)
        assert np.array_equal(O.demod_any_burst(cap, rc1, 4, e1), s0)
    finally:
        O.use_ref_arch(None)


@pytest.mark.skipif(not O.ref_arch_available("sse"), reason="oracle/_ref not built (no /root/reference at build time)")
def test_the_references_sse_path_stays_within_the_north_star_bar():
    """The same with the reference's SSE3 kernels (arch/x86/convolve_sse_3.c, compiled unmodified: libref_sse.so): the oracle
    then IS the reference's SSE path.  Against the generic-C path: rc identical, TOA identical, amp and soft bits within the
    SSE kernels' own summation-order spread -- far inside north star's 1e-4 relative."""
    loads = _workloads()
    try:
        O.use_ref_arch(None)
        base = [O.pull_batch(iq, sps, p, soft_stride=444 if name == "EDGE" else 156, slice_bits=False) for name, iq, p, sps in loads]
        O.use_ref_arch("sse")
        worst = {}
        for (name, iq, p, sps), (r0, sf0) in zip(loads, base):
            r1, sf1 = O.pull_batch(iq, sps, p, soft_stride=444 if name == "EDGE" else 156, slice_bits=False)
            assert np.array_equal(r1["rc"], r0["rc"]) and np.array_equal(r1["tsc"], r0["tsc"]), name
            assert np.array_equal(r1["toa"], r0["toa"]), name
            det = r0["rc"] > 0
            amp0 = np.hypot(r0["amp_re"], r0["amp_im"])[det]
            d_amp = np.hypot(r1["amp_re"] - r0["amp_re"], r1["amp_im"] - r0["amp_im"])[det] / amp0
            assert d_amp.max() < 2e-6, (name, d_amp.max())
            assert np.abs(r1["ci"] - r0["ci"])[det].max() < 1e-3, name
            tol = 5e-5 if name == "EDGE" else 1e-5
            assert np.abs(sf1 - sf0).max() < tol, (name, np.abs(sf1 - sf0).max())
            worst[name] = (float(d_amp.max()), float(np.abs(sf1 - sf0).max()))
        assert max(v[1] for v in worst.values()) > 0          # it really is a different summation order
    finally:
        O.use_ref_arch(None)


def _stage_inputs(v):
    cap = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nb_chunk_tsc7.cfile"), dtype=np.complex64)[:625]
    return [cap] + [b.astype(np.float32).view(np.complex64).reshape(625) for b in v["iq"]]


def test_stage_vectors_of_the_reference_kernels(golden_dir):
    """tests/golden/ref_stage_vectors.npz (tests/golden/make_stage_vectors.py): decimator, correlation at N = 16 / 40 / 64 and the
    20-tap fractional delay of the captured burst and 64 seeded bursts, computed by the reference's unmodified arch objects with
    the arguments of the reference's call sites.  The oracle's own stage functions reproduce every value bit for bit."""
    v = np.load(os.path.join(golden_dir, "ref_stage_vectors.npz"))
    L = O.lib()
    T = O.tables()
    xs = _stage_inputs(v)
    assert len(xs) == 65
    r = L.orc_resampler_new(1, 4, 16, 1.0)
    i_nb = i_ab = 0
    for k, x in enumerate(xs):
        buf = np.concatenate([np.zeros(16, np.complex64), x[:624]])
        dec = np.zeros(156, dtype=np.complex64)
        assert L.orc_resampler_rotate(r, buf[16:].ctypes.data, 624, dec.ctypes.data, 156) == 156
        assert np.array_equal(dec.view(np.float32), v["dec"][k]), k
        if v["kind"][k] == 0:
            c = O.convolve(dec, T["midamble"][int(v["tsc"][k])]["seq"], 71, 19, True)
            assert np.array_equal(c.view(np.float32), v["corr_nb"][i_nb]), k
            i_nb += 1
        else:
            c = O.convolve(dec, T["rach"][0]["seq"], 39, 79, True)
            assert np.array_equal(c.view(np.float32), v["corr_ab"][i_ab]), k
            i_ab += 1
        c = O.convolve(dec, T["sch"]["seq"], 63, 93, True)
        assert np.array_equal(c.view(np.float32), v["corr_sch"][k]), k
        # delayVector with a purely fractional delay that selects filter `filt`: floorf(frac * 64) = filt, no integer shift
        out = np.zeros(625, dtype=np.complex64)
        L.orc_delay_vector(np.ascontiguousarray(x).ctypes.data, 625, float(np.float32((int(v["filt"][k]) + 0.5) / 64.0)), out.ctypes.data)
        got = np.concatenate([out[a:b] for a, b in v["delay_win"]])
        assert np.array_equal(got.view(np.float32), v["delay"][k]), k
    L.orc_resampler_free(r)
    assert i_nb == 33 and i_ab == 32
