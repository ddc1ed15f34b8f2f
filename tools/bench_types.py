#!/usr/bin/env python3
"""Mbursts/s of trxhip_detect_demod_batch on 1M-burst batches by slot type: normal, access, extended access, 7:1 mix (the kernel
split on / off): python tools/bench_types.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from osmo_trx_amd import TrxHip, synth, trxhip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
trx = TrxHip(0)
n = 1 << 20
sets = {"normal": synth.make_normal_bursts(n, "cuda:0", 4)[:2], "rach": synth.make_access_bursts(n, "cuda:0")[:2],
        "ext_rach": synth.make_access_bursts(n, "cuda:0", ext=True)[:2], "mixed": synth.make_mixed_bursts(n, "cuda:0")}
for name, (iq, params) in sets.items():
    d_p = trx.params_tensor(params)
    res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
    soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
    out = []
    hint = trxhip.few_nb_hint(params)
    for split in (True, False):
        trx.set_nb_kernel(split)
        tw = time.perf_counter()
        while time.perf_counter() - tw < 1.0:                       # settle the clocks (the first launches of a process run slow)
            for _ in range(20):
                trx.detect_demod(iq, d_p, sps=4, results=res, soft=soft, hint=hint)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            trx.detect_demod(iq, d_p, sps=4, results=res, soft=soft, hint=hint)
        torch.cuda.synchronize()
        out.append(n * steps / (time.perf_counter() - t0) / 1e6)
    print(f"{name:9s} split {out[0]:8.1f}  general kernel alone {out[1]:8.1f} Mbursts/s", flush=True)
