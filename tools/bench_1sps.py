import os, sys
sys.path.insert(0, os.getcwd())
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
trx = TrxHip(0)
iq1, p1, _ = synth.make_normal_bursts(n, "cuda:0", 1, burst_len=156, tsc=0)
dp1 = trx.params_tensor(p1)
res1 = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
f = lambda: trx.detect_demod(iq1, dp1, sps=1, soft_stride=148, results=res1, soft=soft)
for _ in range(100): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"1 SPS: {ms:.4f} ms {n/ms/1e3:.1f} Mbursts/s")
