"""GPU parity of the Viterbi alternative (cfg->use_va): trxhip_demod_va_batch_cf32 vs the oracle restatement of
scaleVector + demodAnyBurst_va (Transceiver.cpp:782-784, :620-645 over grgsm_vitac/).  The oracle's Viterbi core is
pinned bit for bit to the reference's viterbi_detector.cc compiled unmodified (tests/test_oracle.py); the front end
(channel estimate, matched filter) lives in grgsm_vitac.cpp, which needs libosmocore and is not buildable here.
Bar: burst start and every +-127 output identical."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle import _va_burst
from osmo_trx_amd.trxhip import PARAMS_DTYPE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trx():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from osmo_trx_amd import TrxHip
    return TrxHip(0)


def _run(trx, bursts, params, **kw):
    import torch
    x = torch.from_numpy(np.stack(bursts)).to("cuda:0")
    soft, starts = trx.demod_va(x, trx.params_tensor(params), **kw)
    return soft.cpu().numpy(), starts.cpu().numpy()


def _check(bursts, params, soft, starts, stride=156):
    for b, y in enumerate(bursts):
        st, ref = O.demod_any_burst_va(y, int(params["type"][b]), int(params["tsc"][b]), int(params["max_toa"][b]))
        assert starts[b] == st, (b, starts[b], st)
        assert np.array_equal(soft[b, :156][:stride], ref[:stride]), b


def test_va_normal_bursts_bit_exact(trx):
    rng = np.random.default_rng(41)
    n = 384
    bursts, params = [], np.zeros(n, dtype=PARAMS_DTYPE)
    nerr = 0
    allbits = []
    for b in range(n):
        y, bits = _va_burst(rng, b % 8, int(rng.integers(0, 22)), snr_db=rng.uniform(3, 35), amp=10 ** rng.uniform(2.5, 4.2))
        bursts.append(y)
        allbits.append(bits)
        params[b] = (O.TSC, b % 8, 3, 0)
    soft, starts = _run(trx, bursts, params)
    _check(bursts, params, soft, starts)
    for b in range(n):
        nerr += int(((soft[b, :148] > 0).astype(np.uint8) != allbits[b]).sum())
    assert nerr < 0.05 * n * 148                                       # it demodulates (SNR down to 3 dB included)


def test_va_access_bursts_and_start_state_quirk(trx):
    rng = np.random.default_rng(42)
    ab = np.array([int(c) for c in "01001011011111111001100110101010001111000"], dtype=np.uint8)
    n = 192
    bursts, params = [], np.zeros(n, dtype=PARAMS_DTYPE)
    for b in range(n):
        bits = np.concatenate([np.array([0, 0, 1, 1, 1, 0, 1, 0], np.uint8), ab, rng.integers(0, 2, 36, dtype=np.uint8),
                               np.zeros(3, np.uint8)])
        x = O.modulate_burst(bits, 8, 4)
        off = int(rng.integers(0, 26))
        y = np.zeros(625, dtype=np.complex64)
        m = min(len(x), 625 - off)
        y[off:off + m] = x[:m] * np.complex64(4000.0 * np.exp(1j * rng.uniform(0, 6.28)))
        y += ((rng.normal(size=625) + 1j * rng.normal(size=625)) * rng.uniform(10, 800)).astype(np.complex64)
        bursts.append(y)
        params[b] = (O.RACH if b % 2 else O.EXT_RACH, 0, [0, 3, 12, 15, 16, 63][b % 6], 0)
    soft, starts = _run(trx, bursts, params)
    _check(bursts, params, soft, starts)
    assert not soft[:, 88:].any()


def test_va_noise_lengths_slicer_and_bad_tsc(trx):
    rng = np.random.default_rng(43)
    # pure noise of different scales: every decision is a coin toss decided by rounding -> strict test of operand order
    for L in (600, 625, 665, 1500):
        n = 64
        bursts = [((rng.normal(size=L) + 1j * rng.normal(size=L)) * 10 ** rng.uniform(0, 4)).astype(np.complex64)
                  for _ in range(n)]
        params = np.zeros(n, dtype=PARAMS_DTYPE)
        params["type"] = np.where(np.arange(n) % 3 == 0, O.RACH, O.TSC)
        params["tsc"] = np.arange(n) % 8
        params["max_toa"] = 3
        soft, starts = _run(trx, bursts, params)
        _check(bursts, params, soft, starts)
    sl, _ = _run(trx, bursts, params, soft_stride=148, slice_bits=True)
    assert np.array_equal(sl, (soft[:, :148] > 0).astype(np.float32) * np.where(soft[:, :148] != 0, 1.0, 0.0)
                          + np.where(soft[:, :148] == 0, 0.5, 0.0))   # vectorSlicer: +-127 -> 1 / 0, 0 -> 0.5
    params["tsc"][5] = 9
    soft2, starts2 = _run(trx, bursts, params)
    assert starts2[5] == -1 and not soft2[5].any()
    assert np.array_equal(np.delete(soft2, 5, 0), np.delete(soft, 5, 0))


def test_va_argument_errors(trx):
    import torch
    from osmo_trx_amd.trxhip import TrxHipError
    x = torch.zeros((2, 625), dtype=torch.complex64, device="cuda:0")
    p = trx.params_tensor(np.zeros(2, dtype=PARAMS_DTYPE))
    with pytest.raises(TrxHipError):
        trx.demod_va(x, p, soft_stride=100)
    big = torch.zeros((1, 2000), dtype=torch.complex64, device="cuda:0")
    with pytest.raises(TrxHipError):
        trx.demod_va(big, trx.params_tensor(np.zeros(1, dtype=PARAMS_DTYPE)))
