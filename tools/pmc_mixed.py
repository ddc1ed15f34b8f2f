#!/usr/bin/env python3
"""One workload per launch for counter passes: WORKLOAD=normal|rach|ext|mixed|1sps|edge [EXACT=1] python tools/pmc_mixed.py (4 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
wl = os.environ.get("WORKLOAD", "mixed")
trx = TrxHip(0)
if wl == "mixed":
    iq, p = synth.make_mixed_bursts(n, "cuda:0")
elif wl in ("rach", "ext"):
    iq, p, _ = synth.make_access_bursts(n, "cuda:0", ext=(wl == "ext"))
elif wl == "edge":
    iq, p, _ = synth.make_edge_bursts(n, "cuda:0")
elif wl == "1sps":
    iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 1, burst_len=156, tsc=0)
else:
    iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(p)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
ss = 444 if wl == "edge" else 148
soft = torch.empty((n, ss), dtype=torch.float32, device="cuda:0")
for _ in range(4):
    trx.detect_demod(iq, dp, sps=1 if wl == "1sps" else 4, soft_stride=ss, slice_bits=True, results=res, soft=soft, exact=bool(os.environ.get("EXACT")))
torch.cuda.synchronize()
