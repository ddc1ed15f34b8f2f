#!/usr/bin/env python3
"""One launch of the diagnostic library with a phase-ablation mask, for rocprofv3 --pmc runs
   (instruction counts per phase = differences between masks).
   TRXHIP_LIB=.../libtrxhip_diag.so DIAG_MASK=0x.. rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- python3 tools/pmc_phase.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 17
trx = TrxHip(0)
iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(params)
for m in [int(v, 0) for v in os.environ.get("DIAG_MASKS", "0").split(",")]:
    trx.detect_demod(iq, dp, _diag_mask=m)
    torch.cuda.synchronize()
