"""EDGE (8-PSK) slots through the fused kernel: 1 Mi bursts, 444 soft bits a burst; 60 warm launches, 20 timed."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
trx = TrxHip(0)
iq, p, _ = synth.make_edge_bursts(n, "cuda:0")
dp = trx.params_tensor(p)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 444), dtype=torch.float32, device="cuda:0")
f = lambda: trx.detect_demod(iq, dp, sps=4, soft_stride=444, results=res, soft=soft)
for _ in range(60): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"EDGE: {ms:.4f} ms {n/ms/1e3:.1f} Mbursts/s  {(2500 + 1776 + 32) * n / ms / 1e6 / 8000:.3f} of 8 TB/s")
