"""The normal-burst kernel (csrc/trx_kernel_nb.hip) + leftover list against the general kernel alone (rounds 1-5's path,
trxhip_set_nb_kernel(ctx, 0)): results and soft bits must be BIT-IDENTICAL on every workload -- the split only moves bursts
between kernels that do the same arithmetic.  (Parity of the general kernel against the oracle: tests/test_gpu_parity.py;
both run through the split by default.)"""
import numpy as np
import pytest
import torch

import oracle_lib as O
from osmo_trx_amd import TrxHip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trx():
    t = TrxHip(0)
    yield t
    t.close()


def both(trx, iq, params):
    d_iq, d_p = iq.to("cuda:0"), trx.params_tensor(params)
    trx.set_nb_kernel(True)
    res_a, soft_a = trx.detect_demod(d_iq, d_p, sps=4)
    torch.cuda.synchronize()
    trx.set_nb_kernel(False)
    res_b, soft_b = trx.detect_demod(d_iq, d_p, sps=4)
    torch.cuda.synchronize()
    trx.set_nb_kernel(True)
    return res_a, soft_a, res_b, soft_b


def assert_identical(trx, res_a, soft_a, res_b, soft_b):
    a, b = trx.results_to_numpy(res_a), trx.results_to_numpy(res_b)
    for k in a.dtype.names:
        same = (a[k] == b[k]) | ((a[k] != a[k]) & (b[k] != b[k])) if a[k].dtype.kind == "f" else (a[k] == b[k])
        assert same.all(), (k, np.flatnonzero(~same)[:8], a[k][~same][:8], b[k][~same][:8])
    assert torch.equal(soft_a, soft_b), np.flatnonzero((soft_a != soft_b).any(dim=1).cpu().numpy())[:8]
    # and as bytes: the sign of a zero TOA, the payload of a NaN
    assert torch.equal(res_a, res_b), np.flatnonzero((res_a != res_b).any(dim=1).cpu().numpy())[:8]


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4096, 70001])
def test_normal_bursts_any_batch_size(trx, n):
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=101 + n)
    assert_identical(trx, *both(trx, iq, params))


@pytest.mark.parametrize("max_toa", [0, 3, 17, 30, 32, 33, 63])
def test_window_widths(trx, max_toa):
    """max_toa <= 32 runs in the normal-burst kernel (window of one lane per lag), wider windows are left to the general one;
    delays up to the window so that TOAs beyond the straight-line demodulator's geometry (> 9 symbols) are in the batch"""
    iq, params, _ = synth.make_normal_bursts(6000, "cpu", 4, seed=7 + max_toa, max_toa=max_toa, delay_sym=(0.0, float(max(max_toa, 1))))
    res_a, soft_a, res_b, soft_b = both(trx, iq, params)
    assert_identical(trx, res_a, soft_a, res_b, soft_b)
    r = trx.results_to_numpy(res_a)
    assert (r["rc"] > 0).mean() > 0.85
    if max_toa >= 17:
        assert (r["toa"][r["rc"] > 0] > 9.5).any()


def test_early_bursts_and_mixed_types(trx):
    """negative TOAs (bursts that arrive early), IDLE / OFF slots, access bursts, EDGE slots and a tsc beyond 7 in one batch"""
    n = 16384
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=31, max_toa=5, delay_sym=(-3.0, 5.0))
    iq_r, p_r, _ = synth.make_access_bursts(n // 8, "cpu", seed=32)
    iq[5::8] = iq_r
    params[5::8] = p_r
    params = synth.make_idle_off_mix(params, every=16)
    params["type"][3::64] = O.EDGE
    params["tsc"][9::128] = 9
    res_a, soft_a, res_b, soft_b = both(trx, iq, params)
    assert_identical(trx, res_a, soft_a, res_b, soft_b)
    r = trx.results_to_numpy(res_a)
    nb = (params["type"] == O.TSC) & (params["tsc"] < 8)
    assert (r["toa"][nb & (r["rc"] > 0)] < -0.5).any() and (r["rc"][nb] > 0).mean() > 0.85
    assert (r["rc"][params["type"] == O.RACH] == O.RACH).mean() > 0.9


def test_extreme_inputs(trx):
    """real-only / imaginary-only IQ (the addition-only correlation's guard fails: left to the general kernel), silence,
    saturation, single impulses"""
    n = 4096
    iq, params, _ = synth.make_normal_bursts(n, "cpu", 4, seed=55)
    iq = iq.clone()
    iq[0::8, :, 1] = 0
    iq[1::8, :, 0] = 0
    iq[2::8] = 0
    iq[3::8] = 32767
    iq[4::8] = 0
    iq[4::8, 300, 0] = -32768
    iq[6::8, :, 1] = (iq[6::8, :, 1].to(torch.int32) // 4096).to(torch.int16)
    assert_identical(trx, *both(trx, iq, params))


def test_full_size_normal_bursts_and_mix(trx):
    """BASELINE.json configs[1] and configs[4] at full size (1M bursts on one GPU): bit-identical to the general kernel alone"""
    n = 1 << 20
    iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
    assert_identical(trx, *both(trx, iq, params))
    del iq
    iq, params = synth.make_mixed_bursts(n, "cuda:0")
    assert_identical(trx, *both(trx, iq, params))


def test_concurrent_streams_and_threads(trx):
    """two host threads, each with its own stream, launching the split path on one context: every launch gets its own list"""
    import threading
    iq, params, _ = synth.make_normal_bursts(20000, "cpu", 4, seed=77, max_toa=20, delay_sym=(0.0, 20.0))
    d_iq, d_p = iq.to("cuda:0"), trx.params_tensor(params)
    trx.set_nb_kernel(False)
    ref_res, ref_soft = trx.detect_demod(d_iq, d_p, sps=4)
    torch.cuda.synchronize()
    trx.set_nb_kernel(True)
    out = {}

    def work(k):
        s = torch.cuda.Stream()
        rs = []
        with torch.cuda.stream(s):
            for _ in range(12):
                rs.append(trx.detect_demod(d_iq, d_p, sps=4, stream=s))
        s.synchronize()
        out[k] = rs

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        for res, soft in out[k]:
            assert torch.equal(res, ref_res) and torch.equal(soft, ref_soft)


def test_hint_and_feedback_never_change_results(trx):
    """TRXHIP_FLAG_FEW_NB_SLOTS (host_params given) and the library's own feedback -- a context that was left more than 1/32 of a
    batch runs the general kernel alone for a while -- choose between kernels that give the same bits: an access-burst batch and
    the 7:1 mix, hinted and unhinted, repeated so that the unhinted context passes through probe -> back-off -> probe."""
    n = 65536
    mixed = synth.make_mixed_bursts(n, "cpu")
    rach = synth.make_access_bursts(n, "cpu", seed=91)[:2]
    nb = synth.make_normal_bursts(n, "cpu", 4, seed=92)[:2]
    t2 = TrxHip(0)                                                  # a fresh context: its feedback state starts empty
    try:
        for w, (iq, params) in enumerate((mixed, rach, nb, mixed)):
            d_iq, d_p = iq.to("cuda:0"), t2.params_tensor(params)
            t2.set_nb_kernel(False)
            ref_res, ref_soft = t2.detect_demod(d_iq, d_p, sps=4)
            torch.cuda.synchronize()
            t2.set_nb_kernel(True)
            for k in range(70):                                     # > 4 slots + 63 back-off launches + a probe
                res, soft = t2.detect_demod(d_iq, d_p, sps=4, host_params=params if k % 7 == 3 else None)
                if k % 10 == 0 or k > 66:
                    torch.cuda.synchronize()
                    if not (torch.equal(res, ref_res) and torch.equal(soft, ref_soft)):
                        rows = np.flatnonzero((res != ref_res).any(dim=1).cpu().numpy())
                        srows = np.flatnonzero((soft != ref_soft).any(dim=1).cpu().numpy())
                        raise AssertionError((w, k, len(rows), rows[:8], params["type"][rows[:8]], len(srows), srows[:8],
                                              t2.results_to_numpy(res)[rows[:4]], t2.results_to_numpy(ref_res)[rows[:4]]))
    finally:
        t2.close()
