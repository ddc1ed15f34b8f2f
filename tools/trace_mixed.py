#!/usr/bin/env python3
"""The 7:1 mix (BASELINE.json configs[4], one GPU's share) through the kernel split, for rocprofv3 --kernel-trace --stats."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from osmo_trx_amd import TrxHip, synth
trx = TrxHip(0)
n = 1 << 20
iq, params = synth.make_mixed_bursts(n, "cuda:0")
d_p = trx.params_tensor(params)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    trx.detect_demod(iq, d_p, sps=4, results=res, soft=soft, host_params=None if os.environ.get('NOHINT') else params)
torch.cuda.synchronize()
