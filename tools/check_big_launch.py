#!/usr/bin/env python3
"""One launch of N bursts against the same N bursts as 1M-burst launches (same process, same buffers): records and soft bits must
be bit-identical.  N = 8M exercises what no 1M-burst run does: > 64 pool groups per workgroup (the LDS ring wraps), 21 GB of IQ.
   python tools/check_big_launch.py [n_millions] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from osmo_trx_amd import TrxHip, synth
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
trx = TrxHip(0)
base_iq, base_p = synth.make_mixed_bursts(1 << 16, dev, seed=4321, chunk=4096)
n = nm << 20
# burst b of the big batch = base burst ((b // 8 * 40503) % 8192) * 8 + b % 8: every 8th slot stays an access burst
b = torch.arange(n, device=dev)
idx = ((b // 8 * 40503) % 8192) * 8 + b % 8
iq = base_iq[idx]
params = base_p[idx.cpu().numpy()]
dp = trx.params_tensor(params)
del idx, b
for pool in (True, False):
    trx.set_work_pool(pool)
    for rep in range(reps):
        res = torch.full((n, 32), 0xA5, dtype=torch.uint8, device=dev)
        soft = torch.full((n, 148), float("nan"), dtype=torch.float32, device=dev)
        trx.detect_demod(iq, dp, sps=4, results=res, soft=soft)
        torch.cuda.synchronize()
        bad_total = 0
        for k in range(nm):
            sl = slice(k << 20, (k + 1) << 20)
            r2, s2 = trx.detect_demod(iq[sl], dp[sl], sps=4)
            torch.cuda.synchronize()
            badr = (res[sl] != r2).any(dim=1)
            bads = (soft[sl].view(torch.int32) != s2.view(torch.int32)).any(dim=1)
            bad = badr | bads
            nb = int(bad.sum())
            bad_total += nb
            if nb:
                w = torch.nonzero(bad)[:, 0][:8].cpu().numpy() + (k << 20)
                print(f"  pool={pool} rep={rep} shard {k}: {nb} rows differ (records {int(badr.sum())}, soft {int(bads.sum())}); first {w.tolist()}")
                for i in w[:3]:
                    a = trx.results_to_numpy(res[i:i + 1]); c = trx.results_to_numpy(r2[i - (k << 20):i - (k << 20) + 1])
                    print("    big  ", a[0]); print("    small", c[0], "group", i >> 4, "type", params["type"][i])
        print(f"pool={pool} rep={rep}: {bad_total} of {n} rows differ", flush=True)
