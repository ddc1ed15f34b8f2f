#!/usr/bin/env python3
"""Detect + demod on complex64 bursts (the sigProcLib-signature input and the multi-ARFCN front end's channel streams):
ms per 1M normal bursts after 100 warm launches (same-box A/B: TRXHIP_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
trx = TrxHip(0)
iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4)
cf = torch.view_as_complex(iq.to(torch.float32)).contiguous()
del iq
dp = trx.params_tensor(p)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
f = lambda: trx.detect_demod(cf, dp, sps=4, soft_stride=148, results=res, soft=soft)
for _ in range(100): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"complex64 input: {ms:.4f} ms {n/ms/1e3:.1f} Mbursts/s")
