#!/usr/bin/env python3
"""detectSCHBurst on the GPU: FULL search over 16384 x 625-sample buffers, BUFFER search (12 frames) over 256 x 60000 samples
(same-box A/B: TRXHIP_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip
trx = TrxHip(0)
g = torch.Generator(device="cuda:0"); g.manual_seed(7)
for name, n, L, state in (("FULL", 16384, 625, 0), ("BUFFER", 256, 60000, 2)):
    x = torch.randn((n, L, 2), generator=g, device="cuda:0") * 1000.0
    cf = torch.view_as_complex(x.contiguous())
    f = lambda: trx.detect_sch(cf, state=state)
    for _ in range(30): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"SCH {name}: {ms:.4f} ms per {n} buffers = {n / ms / 1e3:.2f} Mbuffers/s")
