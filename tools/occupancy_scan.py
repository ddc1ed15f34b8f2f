#!/usr/bin/env python3
"""Throughput vs waves per CU (diagnostic build): TRXHIP_LIB=.../libtrxhip_diag.so python tools/occupancy_scan.py"""
import os, sys, subprocess
if len(sys.argv) == 1:
    for w in (1, 2, 4, 6, 8, 10, 12, 14, 16):
        env = dict(os.environ, TRXHIP_WPB=str(w))
        out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip()
        print(f"waves/CU {w:2d}: {out}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
trx = TrxHip(0)
iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(params)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0"); soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
for _ in range(2): trx.detect_demod(iq, dp, results=res, soft=soft)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): trx.detect_demod(iq, dp, results=res, soft=soft)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
w = int(os.environ.get("TRXHIP_WPB", 12))
cyc = ms * 1e-3 * 2.4e9 * 256 * w / n
print(f"{ms:7.3f} ms  {n / ms / 1e3:7.1f} Mbursts/s  ~{cyc:7.0f} cycles per burst per wave (at 2.4 GHz)")
