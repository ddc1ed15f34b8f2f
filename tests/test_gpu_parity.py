"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI (include/trxhip.h), against
the CPU oracle on the same seeded synthetic IQ and against the reference's golden fixtures.

Bars (BASELINE.json north_star: <=1e-4 relative on TOA / RSSI / soft bits):
  * rc, tsc, clip/idle flags, TOA, amp, soft bits: BIT-EXACT vs the generic-C-order oracle (decisions and
    every FIR sum keep the reference's operand order; kernels are built with -ffp-contract=off)
  * energy (tree-summed on the GPU): <= 3e-6 relative;  RSSI (hardware log2): <= 2e-5 dB absolute
  * C/I (device log2f): <= 2e-5 dB absolute
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def trx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from osmo_trx_amd import TrxHip
    return TrxHip(0)


def run_gpu(trx, iq, params, sps, soft_stride=148, slice_bits=True, exact=True, **kw):
    d_iq = iq.to("cuda:0") if not iq.is_cuda else iq
    d_p = trx.params_tensor(params)
    res, soft = trx.detect_demod(d_iq, d_p, sps=sps, soft_stride=soft_stride, slice_bits=slice_bits, exact=exact, **kw)
    torch.cuda.synchronize()
    return trx.results_to_numpy(res), soft.cpu().numpy()


header_constant = O.header_constant
fast_ci_bar = O.fast_ci_bar
FAST_AMP_RTOL = header_constant("TRXHIP_FAST_AMP_RTOL")                 # fused kernels: amp against the reference's


FUSED_SOFT_ATOL = header_constant("TRXHIP_FUSED_SOFT_ATOL")             # fused demodulator, GMSK soft bits (full scale 1)
FUSED_SOFT_ATOL_8PSK = header_constant("TRXHIP_FUSED_SOFT_ATOL_8PSK")   # 8-PSK rows and the fuzz inputs


def assert_same_detection(r, rx):
    """Result records of the fused kernel (FAST detector) against the bit-exact kernel's (or the oracle's): rc, TSC, flags,
    TOA, energy and RSSI bit for bit; amp and C/I inside the bars of include/trxhip.h."""
    for f in ("rc", "tsc", "clip", "idle", "nbits_div4"):
        assert np.array_equal(r[f], rx[f]), f
    for f in ("toa", "energy", "rssi"):
        assert np.array_equal(r[f], rx[f], equal_nan=True), f
    aref = np.hypot(rx["amp_re"], rx["amp_im"])
    assert (np.hypot(r["amp_re"] - rx["amp_re"], r["amp_im"] - rx["amp_im"]) <= FAST_AMP_RTOL * aref).all()
    O.assert_fast_ci(r["ci"], rx["ci"])


def check_parity(g_res, g_soft, o_res, o_soft, soft_atol=0.0):
    """GPU against oracle.  soft_atol = 0: everything bit-identical (exact demodulator).  Otherwise the fused
    demodulator's statement of include/trxhip.h, both clauses: |soft - ref| <= soft_atol * max(1, rms / (4 |amp|)) with
    rms = sqrt(energy) and amp the amplitude estimate -- the plain bar on every real detection, amplitude-aware on noise
    slots detected far below their samples' level -- and identical hard decisions wherever the reference is not within
    10 bars of the decision threshold."""
    for f in ("rc", "tsc", "clip", "idle", "nbits_div4"):
        assert np.array_equal(g_res[f], o_res[f]), f
    assert np.array_equal(g_res["toa"], o_res["toa"])                   # TOA: identical in both kernels
    if soft_atol == 0.0:
        for f in ("amp_re", "amp_im"):
            assert np.array_equal(g_res[f], o_res[f]), f
    else:
        # fused kernels' FAST detector: the interpolated peak is an FMA sum (include/trxhip.h, TRXHIP_FAST_AMP_RTOL)
        aref = np.hypot(o_res["amp_re"], o_res["amp_im"])
        d = np.hypot(g_res["amp_re"] - o_res["amp_re"], g_res["amp_im"] - o_res["amp_im"])
        assert (d <= FAST_AMP_RTOL * aref).all(), float((d / np.maximum(aref, 1e-30)).max())
    if soft_atol == 0.0:
        assert np.array_equal(g_soft, o_soft)
    else:
        amp = np.hypot(o_res["amp_re"], o_res["amp_im"])
        ratio = np.where(amp > 0, np.sqrt(np.maximum(o_res["energy"], 0)) / np.maximum(amp, 1e-30), 1.0)
        bar = (soft_atol * np.maximum(1.0, ratio / 4.0))[:, None]
        err = np.abs(g_soft - o_soft)
        assert (err <= bar).all(), float((err / bar).max())
        sure = np.abs(o_soft - (0.5 if o_soft.min() >= 0 else 0.0)) > 10 * bar
        ref_mid = 0.5 if o_soft.min() >= 0 else 0.0
        assert np.array_equal((g_soft > ref_mid)[sure], (o_soft > ref_mid)[sure])      # same hard decisions
    np.testing.assert_allclose(g_res["energy"], o_res["energy"], rtol=3e-6, atol=0)       # 80-term tree sum vs serial sum
    fin = np.isfinite(o_res["rssi"])
    np.testing.assert_allclose(g_res["rssi"][fin], o_res["rssi"][fin], rtol=0, atol=2e-5)
    assert np.array_equal(np.isfinite(g_res["rssi"]), fin)
    if soft_atol:
        nan = np.isnan(o_res["ci"])                                      # (S < C on a noise slot: log of a negative number, both sides)
        assert np.array_equal(np.isnan(g_res["ci"]), nan)
        assert (np.abs(g_res["ci"] - o_res["ci"])[~nan] <= fast_ci_bar(o_res["ci"][~nan])).all()
    else:
        np.testing.assert_allclose(g_res["ci"], o_res["ci"], rtol=0, atol=2e-5)


def test_normal_bursts_4sps_all_tsc(trx):
    """BASELINE.json configs[1] at an oracle-sized batch: all 8 TSCs, noise-only and clipped bursts included."""
    from osmo_trx_amd import synth
    iq, params, truth = synth.make_normal_bursts(4096, "cpu", 4)
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    g_res, g_soft = run_gpu(trx, iq, params, 4)
    assert (o_res["rc"] > 0).sum() > 3500 and (o_res["rc"] == 0).sum() > 100
    check_parity(g_res, g_soft, o_res, o_soft)
    # default (fused) demodulator: detection bit-exact, soft bits within 1e-5 of full scale
    f_res, f_soft = run_gpu(trx, iq, params, 4, exact=False)
    check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)


@pytest.mark.skipif(not O.ref_arch_available("sse"), reason="oracle/_ref/libref_sse.so not built (it travels prebuilt from the build container)")
def test_against_the_references_sse_path(trx):
    """North star, literally: "outputs (detected TOA, RSSI, soft bits) must match the reference CPU/SSE path to <= 1e-4
    relative".  The oracle's call graph over the reference's own SSE3 kernels (arch/x86/convolve_sse_3.c compiled unmodified
    into oracle/_ref/libref_sse.so, orc_set_arch) IS that path.  Both demodulators against it, normal and access bursts:
    rc, TSC and TOA identical; RSSI within 2e-5 dB; amplitude within 2e-6 relative; soft bits within 2e-5 of full scale (the
    SSE path itself sits up to 3e-6 from the generic-C path, the fused demodulator up to 1e-5: both inside 1e-4)."""
    from osmo_trx_amd import synth
    loads = [synth.make_normal_bursts(4096, "cpu", 4, seed=901)[:2], synth.make_access_bursts(2048, "cpu", seed=902)[:2],
             synth.make_normal_bursts(1024, "cpu", 4, seed=903, max_toa=40, delay_sym=(-2.0, 38.0))[:2]]
    try:
        O.use_ref_arch("sse")
        refs = [O.pull_batch(iq.numpy(), 4, p) for iq, p in loads]
    finally:
        O.use_ref_arch(None)
    for (iq, p), (o_res, o_soft) in zip(loads, refs):
        for exact in (True, False):
            g_res, g_soft = run_gpu(trx, iq, p, 4, exact=exact)
            for f in ("rc", "tsc", "idle", "nbits_div4"):
                assert np.array_equal(g_res[f], o_res[f]), f
            assert np.array_equal(g_res["toa"], o_res["toa"])
            det = o_res["rc"] > 0
            assert det.sum() > 0.85 * len(p)
            amp = np.hypot(o_res["amp_re"], o_res["amp_im"])[det]
            d_amp = np.hypot(g_res["amp_re"] - o_res["amp_re"], g_res["amp_im"] - o_res["amp_im"])[det]
            assert (d_amp / amp).max() < 2e-6
            fin = np.isfinite(o_res["rssi"])
            assert np.abs(g_res["rssi"][fin] - o_res["rssi"][fin]).max() < 2e-5
            # soft bits: absolute on full scale 1, amplitude-aware on noise slots as in check_parity
            ratio = np.where(det, np.sqrt(np.maximum(o_res["energy"], 0)) / np.maximum(np.hypot(o_res["amp_re"], o_res["amp_im"]), 1e-30), 1.0)
            bar = (2e-5 * np.maximum(1.0, ratio / 4.0))[:, None]
            assert (np.abs(g_soft - o_soft) <= bar).all()


def test_normal_bursts_wide_window_raw_soft(trx):
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(1024, "cpu", 4, seed=11, max_toa=63, delay_sym=(-2.0, 40.0))
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=156, slice_bits=False)
    g_res, g_soft = run_gpu(trx, iq, params, 4, soft_stride=156, slice_bits=False)
    assert (o_res["rc"] > 0).sum() > 800
    check_parity(g_res, g_soft, o_res, o_soft)
    # fused demodulator with large and negative TOAs: exercises both truncation edges of delayVector/downsampleBurst
    f_res, f_soft = run_gpu(trx, iq, params, 4, soft_stride=156, slice_bits=False, exact=False)
    check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)


def test_window_widths_around_the_narrow_buffer_limit(trx):
    """The 4-SPS kernel keeps correlation windows up to max_toa = 64 whole in LDS and switches to a windowed
    evaluation beyond (TRX_CORR_NARROW / TRX_DEC_NARROW, trx_device.h).  Per-burst max_toa on both sides of the
    limit, narrow and wide bursts interleaved in one batch (a wide burst must leave the buffers clean for the next),
    NB and RACH, up to the API limit 112: all bit-exact."""
    from osmo_trx_amd import synth
    n = 1536
    widths = np.array([3, 64, 65, 73, 74, 63, 90, 112, 30, 101, 102, 5], dtype=np.uint16)
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=23, max_toa=63, delay_sym=(-1.0, 60.0))
    params["max_toa"] = widths[np.arange(n) % len(widths)]
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    for exact in (True, False):
        g_res, g_soft = run_gpu(trx, iq, params, 4, exact=exact)
        check_parity(g_res, g_soft, o_res, o_soft, soft_atol=0.0 if exact else FUSED_SOFT_ATOL)
    assert (o_res["rc"] > 0).sum() > 900
    wide = params["max_toa"] > 64
    assert ((o_res["rc"] > 0) & wide).sum() > 300 and ((o_res["rc"] > 0) & ~wide).sum() > 300
    iq, params, _ = synth.make_access_bursts(n, "cpu", ext=True, seed=24)
    params["max_toa"] = widths[np.arange(n) % len(widths)]
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    g_res, g_soft = run_gpu(trx, iq, params, 4)
    check_parity(g_res, g_soft, o_res, o_soft)
    assert ((o_res["rc"] > 0) & wide).sum() > 100


@pytest.mark.parametrize("ext", [False, True])
def test_access_bursts(trx, ext):
    """BASELINE.json configs[2]: RACH correlation sweep, max_toa 63; EXT_RACH tries TS0/1/2, first hit wins."""
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_access_bursts(2048, "cpu", ext=ext)
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    g_res, g_soft = run_gpu(trx, iq, params, 4)
    assert (o_res["rc"] > 0).sum() > 1800
    check_parity(g_res, g_soft, o_res, o_soft)
    f_res, f_soft = run_gpu(trx, iq, params, 4, exact=False)          # TOA up to 63 symbols: high-side edge outputs
    check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)


def test_mixed_types_idle_off_edge_fallthrough(trx):
    from osmo_trx_amd import synth
    iq, params = synth.make_mixed_bursts(2048, "cpu")
    params = synth.make_idle_off_mix(params)
    params["type"][5::32] = O.EDGE          # EDGE slots carrying GMSK bursts: falls through to TSC (sigProcLib.cpp:1933-1941)
    params["tsc"][9::64] = 9                # invalid TSC -> -SIGERR_UNSUPPORTED
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
    g_res, g_soft = run_gpu(trx, iq, params, 4)
    assert set(np.unique(o_res["rc"])) >= {-3, 0, 1, 3}
    check_parity(g_res, g_soft, o_res, o_soft)
    f_res, f_soft = run_gpu(trx, iq, params, 4, exact=False)
    check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)


@pytest.mark.parametrize("burst_len", [156, 157])
def test_normal_bursts_1sps_tsc0(trx, burst_len):
    """BASELINE.json configs[0]: 1k normal bursts, 1 SPS, TSC0 (the reference's CPU-runnable case)."""
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(1000, "cpu", 1, tsc=0, amp_range=(8000, 8000), snr_range=(10, 30),
                                             delay_sym=(0, 3), p_noise=0.02, p_clip=0.01, burst_len=burst_len)
    o_res, o_soft = O.pull_batch(iq.numpy(), 1, params)
    g_res, g_soft = run_gpu(trx, iq, params, 1)
    assert (o_res["rc"] > 0).sum() > 950
    check_parity(g_res, g_soft, o_res, o_soft)
    o_res, o_soft = O.pull_batch(iq.numpy(), 1, params, soft_stride=burst_len, slice_bits=False)
    g_res, g_soft = run_gpu(trx, iq, params, 1, soft_stride=burst_len, slice_bits=False)
    check_parity(g_res, g_soft, o_res, o_soft)


def test_captured_burst_golden(trx, golden_dir):
    """The reference's captured burst (utils/va-test) straight through the HIP path as complex64:
    detectAnyBurst(sv, 7, 4.0, 4, TSC, 40) + demodAnyBurst, 1500 samples (burst-gen.cpp:274-290)."""
    x = np.fromfile(os.path.join(golden_dir, "nb_chunk_tsc7.cfile"), dtype=np.complex64)
    bits = np.fromfile(os.path.join(golden_dir, "demodbits_tsc7.s8"), dtype=np.int8)
    params = np.zeros(1, dtype=O.PARAMS_DTYPE)
    params["type"], params["tsc"], params["max_toa"] = O.TSC, 7, 40
    d_iq = torch.from_numpy(x.copy()).view(1, 1500).to("cuda:0")
    g_res, g_soft = run_gpu(trx, d_iq, params, 4, soft_stride=156, slice_bits=False, full_scale=1.0)
    r = g_res[0]
    assert r["rc"] == O.TSC
    assert r["toa"] == np.float32(12.535156)                       # SURVEY.md Appendix A (compiled reference)
    assert abs(r["amp_re"] + 0.00112989) < 1e-8 and abs(r["amp_im"] - 0.00166411) < 1e-8
    assert abs(r["ci"] - 6.460016) < 2e-5
    assert np.array_equal(g_soft[0, :148] > 0, bits > 0)           # 0/148 bit errors
    rc, e = O.detect_any_burst(x, 7, 4.0, 4, O.TSC, 40)
    o_soft = O.demod_any_burst(x, rc, 4, e)
    assert np.array_equal(g_soft[0], o_soft)


def test_full_size_properties(trx):
    """BASELINE.json configs[1] at full size (1M bursts on the device): size-independent properties.
    (a) batch-position independence: a shuffled copy of the batch gives the shuffled results;
    (b) a random sample of the 1M results matches the oracle bit-for-bit;
    (c) detection statistics: every burst with signal is found, no noise-only burst passes with a sane C/I."""
    from osmo_trx_amd import synth
    n = 1 << 20
    iq, params, truth = synth.make_normal_bursts(n, "cuda:0", 4)
    d_p = trx.params_tensor(params)
    res, soft = trx.detect_demod(iq, d_p, sps=4)                         # default = fused demodulator
    perm = torch.randperm(n, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    res2, soft2 = trx.detect_demod(iq[perm].contiguous(), d_p[perm].contiguous(), sps=4)
    res_x, soft_x = trx.detect_demod(iq, d_p, sps=4, exact=True)
    torch.cuda.synchronize()
    assert torch.equal(res[perm], res2) and torch.equal(soft[perm], soft2)
    r = trx.results_to_numpy(res)
    sig = ~truth["noise_only"]
    assert (r["rc"][sig & ~truth["clipped"]] == 1).mean() > 0.9995
    assert (r["rc"][truth["noise_only"]] > 0).mean() < 0.02
    ok = (r["rc"] > 0) & sig
    assert abs(np.mean(r["toa"][ok] - truth["delay_sym"][ok])) < 0.05
    sel = np.random.default_rng(5).choice(n, 2048, replace=False)
    o_res, o_soft = O.pull_batch(iq[torch.from_numpy(sel).to("cuda:0")].cpu().numpy(), 4, params[sel])
    tsel = torch.from_numpy(sel).to("cuda:0")
    check_parity(r[sel], soft[tsel].cpu().numpy(), o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)
    check_parity(trx.results_to_numpy(res_x)[sel], soft_x[tsel].cpu().numpy(), o_res, o_soft)
    # detection decisions identical in both kernels over the whole batch: rc, TSC, flags and TOA bit for bit; amp and C/I
    # of the fused kernel (FAST detector) within the header's bars of the bit-exact kernel's
    assert_same_detection(r, trx.results_to_numpy(res_x))
    assert float((soft - soft_x).abs().max()) <= FUSED_SOFT_ATOL


def test_argument_errors(trx):
    iq = torch.zeros((4, 625, 2), dtype=torch.int16, device="cuda:0")
    p = torch.zeros((4, 8), dtype=torch.uint8, device="cuda:0")
    from osmo_trx_amd import TrxHipError
    with pytest.raises(TrxHipError):
        trx.detect_demod(iq, p, sps=2)
    with pytest.raises(TrxHipError):
        trx.detect_demod(iq[:, :600].contiguous(), p, sps=4)
    # empty batch is a no-op
    trx.detect_demod(iq[:0], p[:0], sps=4)


def test_full_size_access_bursts(trx):
    """BASELINE.json configs[2] at full size: 1M access bursts, max_toa 63 (79-lag, 40-tap correlation sweep)."""
    from osmo_trx_amd import synth
    n = 1 << 20
    iq, params, truth = synth.make_access_bursts(n, "cuda:0")
    d_p = trx.params_tensor(params)
    res, soft = trx.detect_demod(iq, d_p, sps=4)
    torch.cuda.synchronize()
    r = trx.results_to_numpy(res)
    sig = ~truth["noise_only"]
    assert (r["rc"][sig] == O.RACH).mean() > 0.9995
    assert (r["rc"][~sig] > 0).mean() < 0.02
    ok = (r["rc"] > 0) & sig
    assert abs(np.mean(r["toa"][ok] - truth["delay_sym"][ok])) < 0.05
    assert np.std(r["toa"][ok] - truth["delay_sym"][ok]) < 0.1
    # idempotence / batch-position independence on a slice processed alone
    sl = slice(777_000, 777_000 + 4096)
    res2, soft2 = trx.detect_demod(iq[sl].contiguous(), d_p[sl].contiguous(), sps=4)
    assert torch.equal(res[sl], res2) and torch.equal(soft[sl], soft2)
    sel = np.random.default_rng(9).choice(n, 1024, replace=False)
    tsel = torch.from_numpy(sel).to("cuda:0")
    o_res, o_soft = O.pull_batch(iq[tsel].cpu().numpy(), 4, params[sel])
    check_parity(r[sel], soft[tsel].cpu().numpy(), o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)


def test_edge_8psk_bursts(trx):
    """EDGE slots -- detectEdgeBurst (:1906-1924) + demodEdgeBurst (:2105-2128): 5-tap equaliser,
    3pi/8 derotation, EVM C/I, Manhattan soft slicing to 444 soft bits.  No reference fixture pins 8-PSK: the oracle's
    restatement is the checker ("parity unpinned" for this branch)."""
    from osmo_trx_amd import synth
    iq, params, bits = synth.make_edge_bursts(1024, "cpu")
    params["type"][7::16] = O.TSC            # GMSK detection attempted on an 8-PSK burst
    params["max_toa"][1::5] = 33             # the widest window of the kernel's straight-line EDGE branch ...
    params["max_toa"][2::5] = 34             # ... and the first one that takes the general candidate loop
    for slice_bits in (False, True):
        o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=444, slice_bits=slice_bits)
        assert (o_res["rc"] == O.EDGE).sum() > 900 and (o_res["nbits_div4"] == 111).sum() > 900
        g_res, g_soft = run_gpu(trx, iq, params, 4, soft_stride=444, slice_bits=slice_bits, exact=True)
        check_parity(g_res, g_soft, o_res, o_soft)
        f_res, f_soft = run_gpu(trx, iq, params, 4, soft_stride=444, slice_bits=slice_bits, exact=False)
        check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL_8PSK)     # equaliser gain ~2 on top of the fused demod
    det = o_res["rc"] == O.EDGE
    mid = 0.5
    assert ((g_soft[det] > mid).astype(np.uint8) != bits[det]).mean() < 0.03    # static equaliser: ~1 % raw BER
    # generic kernel (complex64 input, L = 700 > 628) runs the same branch
    cf = torch.view_as_complex(iq[:64].to(torch.float32)).contiguous()
    cf700 = torch.zeros((64, 700), dtype=torch.complex64)
    cf700[:, :625] = cf
    o2_res = np.zeros(64, dtype=O.RESULT_DTYPE)
    o2_soft = np.zeros((64, 444), dtype=np.float32)
    for i in range(64):
        x = cf700[i].numpy()
        rc, e = O.detect_any_burst(x, int(params["tsc"][i]), 4.0, 4, int(params["type"][i]), int(params["max_toa"][i]))
        o2_res["rc"][i] = rc
        if rc > 0:
            s = O.demod_any_burst(x, rc, 4, e)
            o2_soft[i, :len(s)] = s
            o2_res["toa"][i], o2_res["ci"][i] = e.toa, e.ci
    g2_res, g2_soft = run_gpu(trx, cf700, params[:64], 4, soft_stride=444, slice_bits=False, full_scale=1.0)
    assert np.array_equal(g2_res["rc"], o2_res["rc"]) and np.array_equal(g2_res["toa"], o2_res["toa"])
    assert np.array_equal(g2_soft, o2_soft)
    np.testing.assert_allclose(g2_res["ci"], o2_res["ci"], atol=1e-4)


def _fuzz_batch(n, L, rng):
    """Adversarial mix: random slot types / TSCs / search windows over bursts with wild timing, silence,
    full-scale saturation, DC, and single impulses."""
    from osmo_trx_amd import synth
    iq_nb, p_nb, _ = synth.make_normal_bursts(n, "cpu", 4, seed=int(rng.integers(1 << 30)), max_toa=30,
                                              delay_sym=(-9.0, 34.0), p_noise=0.1, p_clip=0.05)
    iq_rb, p_rb, _ = synth.make_access_bursts(n, "cpu", seed=int(rng.integers(1 << 30)), ext=True)
    iq = iq_nb.clone()
    pick = torch.from_numpy(rng.random(n) < 0.35)
    iq[pick] = iq_rb[pick]
    params = p_nb.copy()
    params["type"] = rng.choice([O.OFF, O.TSC, O.EXT_RACH, O.RACH, O.SCH, O.EDGE, O.IDLE, 9], size=n,
                                p=[0.04, 0.4, 0.12, 0.2, 0.02, 0.15, 0.05, 0.02])
    params["tsc"] = rng.integers(0, 9, size=n)                      # 8 is invalid
    params["max_toa"] = rng.choice([0, 1, 3, 30, 63, 100, 112, 113, 400], size=n)
    special = rng.random(n)
    iq[torch.from_numpy(special < 0.02)] = 0                         # silence: energy 0, rssi inf
    iq[torch.from_numpy((special >= 0.02) & (special < 0.04))] = 32767
    iq[torch.from_numpy((special >= 0.04) & (special < 0.05))] = -32768
    imp = np.where((special >= 0.05) & (special < 0.07))[0]
    for i in imp:
        iq[i] = 0
        iq[i, int(rng.integers(0, 625)), 0] = 20000
    if L != 625:
        out = torch.zeros((n, L, 2), dtype=torch.int16)
        m = min(L, 625)
        out[:, :m] = iq[:, :m]
        iq = out
    return iq, params


@pytest.mark.parametrize("L,soft_stride", [(625, 148), (625, 444), (624, 156), (628, 100), (625, 1), (640, 148), (700, 200)])
def test_fuzz_against_oracle(trx, L, soft_stride):
    """Random slot types, invalid TSCs, every search-window size (incl. beyond TRXHIP_MAX_TOA), TOAs from -9 to +63
    symbols, silence / saturation / impulses, burst lengths either side of the fast kernel's range, odd soft strides."""
    rng = np.random.default_rng(1000 * L + soft_stride)
    iq, params = _fuzz_batch(768, L, rng)
    big = params["max_toa"] > 112
    for slice_bits in (True, False):
        o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=soft_stride, slice_bits=slice_bits)
        # documented limit: max_toa > TRXHIP_MAX_TOA is rejected with -SIGERR_UNSUPPORTED instead of searched
        live = big & ~np.isin(params["type"], [O.OFF, O.IDLE, O.SCH, 9]) & ~((params["tsc"] > 7) & np.isin(params["type"], [O.TSC, O.EDGE]))
        o_res["rc"][live] = -O.SIGERR_UNSUPPORTED
        for f in ("toa", "amp_re", "amp_im", "ci", "tsc", "nbits_div4"):
            o_res[f][live] = 0
        o_res["idle"][live] = 1
        o_soft[live] = 0
        g_res, g_soft = run_gpu(trx, iq, params, 4, soft_stride=soft_stride, slice_bits=slice_bits, exact=True)
        check_parity(g_res, g_soft, o_res, o_soft)
        f_res, f_soft = run_gpu(trx, iq, params, 4, soft_stride=soft_stride, slice_bits=slice_bits, exact=False)
        check_parity(f_res, f_soft, o_res, o_soft, soft_atol=FUSED_SOFT_ATOL_8PSK)


def test_no_soft_output_pointer(trx):
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(256, "cpu", 4, seed=3)
    d_iq, d_p = iq.to("cuda:0"), trx.params_tensor(params)
    res, _ = trx.detect_demod(d_iq, d_p, sps=4)
    res2, soft2 = trx.detect_demod(d_iq, d_p, sps=4, want_soft=False)
    torch.cuda.synchronize()
    assert soft2 is None and torch.equal(res, res2)


def test_capturable_in_a_hip_graph(trx):
    """The batched entry point makes no synchronising call: it can be captured into a HIP graph on a side stream and
    replayed (tools/bench_latency.py measures when that pays)."""
    import torch
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(256, "cuda:0", 4, seed=77)
    dp = trx.params_tensor(params)
    res = torch.empty((256, 32), dtype=torch.uint8, device="cuda:0")
    soft = torch.empty((256, 148), dtype=torch.float32, device="cuda:0")
    trx.detect_demod(iq, dp, results=res, soft=soft)
    torch.cuda.synchronize()
    ref_res, ref_soft = res.clone(), soft.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            trx.detect_demod(iq, dp, results=res, soft=soft)
    res.zero_()
    soft.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(res, ref_res) and torch.equal(soft, ref_soft)


def test_dummy_burst_detection_on_idle_slots(trx):
    """detectAnyBurst(IDLE) -> detectDummyBurst (sigProcLib.cpp:1863-1877, :1945-1947): with TRXHIP_FLAG_IDLE_DUMMY the
    IDLE slots are correlated with gDummySequence (rc = IDLE on a hit) instead of being skipped the way
    pullRadioVector skips them.  Bursts carrying the dummy midamble, a normal TSC, or noise; vs the oracle per burst."""
    import torch
    from osmo_trx_amd import synth
    n = 512
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=55, max_toa=5)
    # re-modulate a third of the bursts with the dummy midamble 0111000101110001011100010 1 (GSM 05.02 5.2.6)
    dummy = np.array([int(c) for c in "01110001011100010111000101"], dtype=np.uint8)
    rng = np.random.default_rng(56)
    x = iq.numpy().copy()
    for b in range(0, n, 3):
        bits = rng.integers(0, 2, 148, dtype=np.uint8)
        bits[:3] = 0; bits[-3:] = 0; bits[61:87] = dummy
        m = O.modulate_burst(bits, 8, 4)
        y = np.zeros(625, dtype=np.complex64)
        off = int(rng.integers(0, 8))
        y[off:off + min(len(m), 625 - off)] = m[:625 - off] * np.complex64(6000 * np.exp(1j * rng.uniform(0, 6.28)))
        y += ((rng.normal(size=625) + 1j * rng.normal(size=625)) * 200).astype(np.complex64)
        x[b, :, 0] = np.clip(np.rint(y.real), -32768, 32767)
        x[b, :, 1] = np.clip(np.rint(y.imag), -32768, 32767)
    iq = torch.from_numpy(x)
    params["type"] = O.IDLE
    # the dummy midamble repeats every 8 bits: its correlation peak-to-neighbourhood ratio stays below BURST_THRESH = 4
    # on clean bursts (the reference would not find them either), so the comparison runs at a threshold of 1.5
    thr = 1.5
    res, soft = trx.detect_demod(iq.to("cuda:0"), trx.params_tensor(params), sps=4, threshold=thr, soft_stride=156,
                                 slice_bits=False, exact=True, idle_dummy=True)
    g = trx.results_to_numpy(res)
    gs = soft.cpu().numpy()
    nhit = 0
    for b in range(n):
        xb = x[b].astype(np.float32).view(np.complex64).reshape(625)
        rc, ebp = O.detect_any_burst(xb, int(params["tsc"][b]), thr, 4, O.IDLE, int(params["max_toa"][b]))
        assert g["rc"][b] == rc, (b, g["rc"][b], rc)
        if rc > 0:
            nhit += 1
            assert rc == O.IDLE and g["toa"][b] == np.float32(ebp.toa) and g["tsc"][b] == 0 and g["idle"][b] == 0
            assert g["amp_re"][b] == np.float32(ebp.amp[0]) and g["amp_im"][b] == np.float32(ebp.amp[1])
            assert np.array_equal(gs[b], O.demod_any_burst(xb, rc, 4, ebp)[:156])
        else:
            assert g["idle"][b] == 1 and not gs[b].any()
    assert nhit >= n // 3 - 8
    # without the flag IDLE slots are skipped (Transceiver.cpp:754-755)
    res2, _ = trx.detect_demod(iq.to("cuda:0"), trx.params_tensor(params), sps=4)
    assert (trx.results_to_numpy(res2)["rc"] == 0).all()


def test_unit_correlation_guard_and_fallback(trx):
    """corr_unit() (trx_device.h) replaces the 4-operation complex MAC by one exact addition per tap when no decimated
    sample has a component more than 2^17 times the other; bursts that break that guard must take the multiplying
    path and still be bit-identical.  Real-only IQ (Q = 0: every decimated sample has an exactly zero imaginary part),
    imaginary-only IQ, a burst with a single tiny Q, and ordinary bursts interleaved in one batch, NB and RACH."""
    from osmo_trx_amd import synth
    n = 1024
    for make in (synth.make_normal_bursts, synth.make_access_bursts):
        iq, params, _ = make(n, "cpu", seed=77) if make is synth.make_access_bursts else make(n, "cpu", 4, seed=77)
        iq = iq.clone()
        iq[0::4, :, 1] = 0                      # real only
        iq[1::4, :, 0] = 0                      # imaginary only
        iq[2::4, :, 1] = (iq[2::4, :, 1].to(torch.int32) // 4096).to(torch.int16)    # Q = -8..7: ratios up to 2^15 .. inf
        o_res, o_soft = O.pull_batch(iq.numpy(), 4, params)
        g_res, g_soft = run_gpu(trx, iq, params, 4)
        check_parity(g_res, g_soft, o_res, o_soft)
        assert (o_res["rc"][3::4] > 0).sum() > 200          # the untouched quarter detects as usual


def test_full_size_mixed(trx):
    """BASELINE.json configs[4], one GPU's share at full size: 1M bursts, 7:1 NB:RACH interleaved (every 8th slot an access
    burst, RACH max_toa 63), through the straight-line NB / RACH paths of the 4-SPS kernel.  Size-independent properties
    (detection statistics per type, TOA statistics, batch-position independence) + a 2048-burst sample against the oracle."""
    from osmo_trx_amd import synth
    n = 1 << 20
    iq, params = synth.make_mixed_bursts(n, "cuda:0")
    d_p = trx.params_tensor(params)
    res, soft = trx.detect_demod(iq, d_p, sps=4)
    res_x, soft_x = trx.detect_demod(iq, d_p, sps=4, exact=True)
    torch.cuda.synchronize()
    r = trx.results_to_numpy(res)
    is_rach = params["type"] == O.RACH
    assert is_rach.sum() == n // 8 and (params["type"][~is_rach] == O.TSC).all()
    # 5 % of either kind are noise-only, 1 % of the normal bursts clipped: detections carry the slot's own type
    assert 0.94 < (r["rc"][is_rach] == O.RACH).mean() < 0.96 and 0.93 < (r["rc"][~is_rach] == O.TSC).mean() < 0.96
    assert not ((r["rc"][is_rach] > 0) & (r["rc"][is_rach] != O.RACH)).any()
    assert not ((r["rc"][~is_rach] > 0) & (r["rc"][~is_rach] != O.TSC)).any()
    det = r["rc"] > 0
    assert (r["nbits_div4"][det] == 37).all() and (r["idle"][det] == 0).all() and (r["idle"][~det] == 1).all()
    # TOA stays inside the search window (:1683 edge gate, "- head"): NB peak index 3..16 minus head 10, RACH 3..76 minus 8
    assert -7.6 < r["toa"][det & ~is_rach].min() and r["toa"][det & ~is_rach].max() < 6.6
    assert -5.6 < r["toa"][det & is_rach].min() and r["toa"][det & is_rach].max() < 68.6
    assert np.median(r["toa"][det & ~is_rach]) < 4.0 and 20.0 < np.median(r["toa"][det & is_rach]) < 45.0
    assert_same_detection(r, trx.results_to_numpy(res_x))
    assert float((soft - soft_x).abs().max()) <= FUSED_SOFT_ATOL
    # position independence: an unaligned slice processed alone (different wave <-> burst assignment)
    sl = slice(123_457, 123_457 + 8191)
    res2, soft2 = trx.detect_demod(iq[sl].contiguous(), d_p[sl].contiguous(), sps=4)
    assert torch.equal(res[sl], res2) and torch.equal(soft[sl], soft2)
    sel = np.sort(np.random.default_rng(11).choice(n, 2048, replace=False))
    tsel = torch.from_numpy(sel).to("cuda:0")
    o_res, o_soft = O.pull_batch(iq[tsel].cpu().numpy(), 4, params[sel])
    check_parity(r[sel], soft[tsel].cpu().numpy(), o_res, o_soft, soft_atol=FUSED_SOFT_ATOL)
    check_parity(trx.results_to_numpy(res_x)[sel], soft_x[tsel].cpu().numpy(), o_res, o_soft)


@pytest.mark.parametrize("stride", [64, 100, 130, 200])
def test_soft_rows_shorter_and_longer_than_a_burst(trx, stride):
    """include/trxhip.h: "the row is truncated to soft_stride"; unused tail zero-filled.  Both demodulators."""
    from osmo_trx_amd import synth
    iq, params, _ = synth.make_normal_bursts(512, "cpu", 4, seed=91)
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=stride)
    for exact in (True, False):
        g_res, g_soft = run_gpu(trx, iq, params, 4, soft_stride=stride, exact=exact)
        check_parity(g_res, g_soft, o_res, o_soft, soft_atol=0.0 if exact else FUSED_SOFT_ATOL)
    assert not o_soft[:, 148:].any()


@pytest.mark.parametrize("sps", [4, 1])
def test_results_do_not_depend_on_the_batch_size(trx, sps):
    """The kernels hand bursts out dynamically (waves claim them from a per-workgroup counter, in groups of 16 consecutive
    bursts per workgroup; soft bits and result record of a burst are stored while the next one is being processed).  None of
    that may show: every burst's record and soft row are the same whether it is processed alone, in a ragged batch (sizes
    around the group and workgroup granularities: 1, 15, 16, 17, 255, 257, 4095, 4097) or in the whole batch, at any offset.
    Mixed slot types so that the different code paths follow each other inside one wave."""
    from osmo_trx_amd import synth
    n = 12288
    if sps == 4:
        iq, params = synth.make_mixed_bursts(n, "cuda:0")
        params["type"][5::64] = O.OFF
        params["type"][9::64] = O.IDLE
    else:
        iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 1)
    d_p = trx.params_tensor(params)
    for exact in (False, True):
        if sps == 1 and not exact:
            continue                                           # one demodulator at 1 SPS
        res, soft = trx.detect_demod(iq, d_p, sps=sps, exact=exact)
        torch.cuda.synchronize()
        off = 0
        for m in (1, 15, 16, 17, 255, 257, 4095, 4097, 1, 2, 33):
            sl = slice(off, off + m)
            r2, s2 = trx.detect_demod(iq[sl].contiguous(), d_p[sl].contiguous(), sps=sps, exact=exact)
            assert torch.equal(res[sl], r2) and torch.equal(soft[sl], s2), (sps, exact, m, off)
            off += m
        assert off <= n


def test_cross_die_pool_equals_static_split(trx):
    """Large batches deal 7/8 of the 16-burst groups statically and let the workgroups draw the rest from a device-wide
    counter (trx_kernel4.hip, cross-die pool).  Which CU processes a burst must not matter: for batch sizes around the
    pool's threshold and with ragged tails, pooled and static (TRXHIP_NO_POOL) launches give bit-identical records and soft
    bits, no burst skipped or done twice (every record of a sentinel-filled output is written), also for back-to-back
    launches on two streams (each launch has its own counter)."""
    from osmo_trx_amd import synth
    n_max = 3 * 65536 + 77
    iq, params = synth.make_mixed_bursts(n_max, "cuda:0", seed=808, chunk=8192)
    d_p = trx.params_tensor(params)
    for n in (32768 - 1, 32768, 65536 + 15, 131072 + 1, n_max):
        outs = []
        for no_pool in (True, False):
            trx.set_work_pool(not no_pool)
            res = torch.full((n, 32), 0xA5, dtype=torch.uint8, device="cuda:0")
            soft = torch.full((n, 148), float("nan"), dtype=torch.float32, device="cuda:0")
            for exact in (False, True):
                trx.detect_demod(iq[:n], d_p[:n], sps=4, soft_stride=148, slice_bits=True, results=res, soft=soft, exact=exact)
                torch.cuda.synchronize()
                outs.append((res.cpu().numpy().copy(), soft.cpu().numpy().copy()))
        assert not np.isnan(outs[2][1]).any() and not np.isnan(outs[3][1]).any()
        for k in (0, 1):
            assert np.array_equal(outs[k][0], outs[k + 2][0]), (n, k)
            assert np.array_equal(outs[k][1].view(np.uint32), outs[k + 2][1].view(np.uint32)), (n, k)
    # many launches at random sizes (pooled and not): always terminates, always the same records as the full-batch run
    full_res, _ = trx.detect_demod(iq, d_p, sps=4)
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    for n in rng.integers(1, n_max + 1, 150).tolist() + [n_max, 32768 + 16, 65536]:
        res, _ = trx.detect_demod(iq[:n], d_p[:n], sps=4)
        assert torch.equal(res, full_res[:n]), n
    # two streams, launches interleaved: every launch draws from its own counter
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ref_res, ref_soft = trx.detect_demod(iq, d_p, sps=4)
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(trx.detect_demod(iq, d_p, sps=4, stream=s))
    torch.cuda.synchronize()
    for r, so in outs:
        assert torch.equal(r, ref_res) and torch.equal(so.view(torch.int32), ref_soft.view(torch.int32))
    # many streams, two launches on each, all outstanding together: every launch has its own counter pair (1024 per context,
    # re-armed by the kernel itself; rounds 3 / 4 shared counters modulo 64 / ran out of per-stream slots after 64 streams)
    streams = [torch.cuda.Stream() for _ in range(70)]
    n = 32768 + 16
    outs = []
    for rep in range(2):
        for s in streams:
            with torch.cuda.stream(s):
                outs.append(trx.detect_demod(iq[:n], d_p[:n], sps=4, stream=s))
    torch.cuda.synchronize()
    for r, so in outs:
        assert torch.equal(r, ref_res[:n]) and torch.equal(so.view(torch.int32), ref_soft[:n].view(torch.int32))


def _oracle_threaded(iq_np, params, threads):
    """O.pull_batch over contiguous slices on `threads` host threads (ctypes releases the GIL; bursts are independent)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(iq_np)
    edges = np.linspace(0, n, threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda k: O.pull_batch(iq_np[edges[k]:edges[k + 1]], 4, params[edges[k]:edges[k + 1]]), range(threads)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def test_fast_detector_campaign_against_oracle(trx, capsys):
    """Round 5's FAST detector (fused kernels: FMA interpolation rounds, tree-summed gate / C/I terms, every early / late
    decision certified by a proven margin or re-run in the reference's operand order): over 4M normal bursts, 1M access bursts
    and the fuzz set, rc / TSC / flags / TOA must be IDENTICAL to the oracle's, amp and C/I inside the header's bars; the
    re-run rate is reported (profiles/r05_fast_campaign.txt holds the 1M-per-workload table of tools/fast_detect_report.py)."""
    from osmo_trx_amd import synth
    import bench
    threads = max(1, min(32, bench.usable_cores()))
    n = 1 << 20
    O.lib()
    total = reruns = detected = 0
    worst_amp = worst_ci = 0.0
    loads = [("nb", k) for k in range(4)] + [("rach", 0)]
    for wl, k in loads:
        if wl == "nb":
            iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4, seed=synth.SEED + 104729 * (k + 1))
        else:
            iq, params, _ = synth.make_access_bursts(n, "cuda:0", seed=synth.SEED + 15485863)
        dp = trx.params_tensor(params)
        trx.fast_stats(reset=True)
        res, soft = trx.detect_demod(iq, dp, sps=4, exact=False)
        st = trx.fast_stats(reset=True)
        g = trx.results_to_numpy(res)
        o_res, o_soft = _oracle_threaded(iq.cpu().numpy(), params, threads)
        for f in ("rc", "tsc", "clip", "idle", "nbits_div4", "toa"):
            assert np.array_equal(g[f], o_res[f]), (wl, k, f)
        det = o_res["rc"] > 0
        aref = np.hypot(o_res["amp_re"], o_res["amp_im"])
        d = np.hypot(g["amp_re"] - o_res["amp_re"], g["amp_im"] - o_res["amp_im"])
        assert (d <= FAST_AMP_RTOL * aref).all()
        ok = det & ~np.isnan(o_res["ci"])
        assert np.array_equal(np.isnan(g["ci"]), np.isnan(o_res["ci"]))
        cie = np.abs(g["ci"] - o_res["ci"])[ok]
        assert (cie <= fast_ci_bar(o_res["ci"][ok])).all()
        worst_amp = max(worst_amp, float((d[det] / aref[det]).max()))
        worst_ci = max(worst_ci, float((cie / fast_ci_bar(o_res["ci"][ok])).max()))
        total += n
        reruns += st["reruns"]
        detected += int(det.sum())
        del iq, res, soft, o_soft
    # the fuzz set (random slot types, invalid TSCs, all window sizes, silence / saturation / impulses): decisions identical
    rng = np.random.default_rng(20261003)
    iq, params = _fuzz_batch(4096, 625, rng)
    o_res, _ = O.pull_batch(iq.numpy(), 4, params)
    big = (params["max_toa"] > 112) & ~np.isin(params["type"], [O.OFF, O.IDLE, O.SCH, 9]) & \
        ~((params["tsc"] > 7) & np.isin(params["type"], [O.TSC, O.EDGE]))
    trx.fast_stats(reset=True)
    g, _ = run_gpu(trx, iq, params, 4, exact=False)
    st = trx.fast_stats(reset=True)
    for f in ("rc", "tsc", "clip", "idle", "nbits_div4", "toa"):
        assert np.array_equal(g[f][~big], o_res[f][~big]), ("fuzz", f)
    total += len(params)
    reruns += st["reruns"]
    rate = reruns / max(1, detected)
    with capsys.disabled():
        print(f"\n[fast detector] {total} bursts, {detected} detected (without the fuzz set): rc / TSC / TOA identical to the oracle; "
              f"{reruns} TOA searches re-run in the reference's operand order ({rate:.3%} of detected); "
              f"max amp error {worst_amp:.2e} relative (bar {FAST_AMP_RTOL:g}), max C/I error {worst_ci:.3f} of its bar")
    assert rate < 0.02


def test_two_host_threads_launching_on_one_stream(trx):
    """the pool counter of round 4 was zeroed by a hipMemsetAsync queued separately from the launch, so two host
    threads launching large batches on ONE stream could enqueue memset A, memset B, kernel A, kernel B and leave kernel B
    with an exhausted counter -- the last eighth of its groups never processed, stale rows returned.  The kernel now re-arms
    its counter pair itself and a launch is a single enqueue: any interleaving of the two threads' calls is correct.  Two
    threads, 24 pooled launches each on the NULL stream, sentinel-filled outputs, every result equal to the serial run."""
    import threading
    from osmo_trx_amd import synth
    n = 65536 + 48
    iq, params = synth.make_mixed_bursts(n, "cuda:0", seed=909, chunk=8192)
    d_p = trx.params_tensor(params)
    ref_res, ref_soft = trx.detect_demod(iq, d_p, sps=4)
    torch.cuda.synchronize()
    outs = {0: [], 1: []}
    bufs = {t: [(torch.full((n, 32), 0xA5, dtype=torch.uint8, device="cuda:0"),
                 torch.full((n, 148), float("nan"), dtype=torch.float32, device="cuda:0")) for _ in range(24)] for t in (0, 1)}
    torch.cuda.synchronize()
    go = threading.Barrier(2)

    def work(t):
        go.wait()
        for res, soft in bufs[t]:
            outs[t].append(trx.detect_demod(iq, d_p, sps=4, results=res, soft=soft))

    th = [threading.Thread(target=work, args=(t,)) for t in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    for t in (0, 1):
        assert len(outs[t]) == 24
        for r, so in outs[t]:
            assert torch.equal(r, ref_res) and torch.equal(so.view(torch.int32), ref_soft.view(torch.int32))
