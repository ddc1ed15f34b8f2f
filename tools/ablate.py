#!/usr/bin/env python3
"""Phase-ablation timing of burst_pull_kernel (diagnostic build -DTRX_DIAG only).
   TRXHIP_LIB=osmo_trx_amd/lib/libtrxhip_diag.so python tools/ablate.py
bits: 0 skip demod | 1 skip TOA bisection | 2 skip log10 | 3 skip detect | 4 skip delay FIR | 5 skip soft epilogue
      6 skip edge rounds | 7 skip C/I | 9 stop detect after the arg-max"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth

n = int(os.environ.get("N", 1 << 19))
trx = TrxHip(0)
iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(params)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
masks = [0, 1, 2, 4, 8, 32, 64, 128, 512, 2 | 128, 1 | 512, 1 | 2 | 128]
for m in masks:
    for _ in range(2):
        trx.detect_demod(iq, dp, results=res, soft=soft, _diag_mask=m)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        trx.detect_demod(iq, dp, results=res, soft=soft, _diag_mask=m)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"mask {m:4d} ({m:010b}): {ms:7.3f} ms  {n / ms / 1e3:8.1f} Mbursts/s")
