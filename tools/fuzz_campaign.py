#!/usr/bin/env python3
"""Longer randomised GPU-vs-oracle campaign than the test suite runs (tests/test_gpu_parity.py::_fuzz_batch inputs,
many seeds): detection fields and exact-demodulator soft bits bit-exact, fused soft bits within 5e-5.
   python tools/fuzz_campaign.py [n_seeds] [bursts_per_seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from osmo_trx_amd import TrxHip
from test_gpu_parity import _fuzz_batch, run_gpu, check_parity

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
trx = TrxHip(0)
tot = det = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(0xF0220000 + seed)
    L = int(rng.choice([625, 625, 625, 624, 626, 628]))
    ss = int(rng.choice([148, 156, 444]))
    sl = bool(rng.integers(0, 2))
    if seed % 2 == 0:                              # every other seed in the geometry of the COMMON instantiation (the product's hot kernel)
        L, ss, sl = 625, 148, True
    iq, params = _fuzz_batch(n, L, rng)
    o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=ss, slice_bits=sl)
    big = params["max_toa"] > 112
    live = big & ~np.isin(params["type"], [O.OFF, O.IDLE, O.SCH, 9]) & ~((params["tsc"] > 7) & np.isin(params["type"], [O.TSC, O.EDGE]))
    o_res["rc"][live] = -O.SIGERR_UNSUPPORTED
    for f in ("toa", "amp_re", "amp_im", "ci", "tsc", "nbits_div4"):
        o_res[f][live] = 0
    o_res["idle"][live] = 1
    o_soft[live] = 0
    g_res, g_soft = run_gpu(trx, iq, params, 4, soft_stride=ss, slice_bits=sl, exact=True)
    check_parity(g_res, g_soft, o_res, o_soft)
    f_res, f_soft = run_gpu(trx, iq, params, 4, soft_stride=ss, slice_bits=sl, exact=False)
    check_parity(f_res, f_soft, o_res, o_soft, soft_atol=5e-5)
    tot += n; det += int((o_res["rc"] > 0).sum())
    print(f"seed {seed:3d}: L {L} stride {ss} slice {int(sl)}  ok  ({int((o_res['rc'] > 0).sum())} detected)", flush=True)
print(f"campaign ok: {tot} bursts, {det} detected, all fields within the parity bars")
