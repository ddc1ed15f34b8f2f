#!/usr/bin/env python3
"""Throughput of the hot kernel on the other BASELINE.json configurations (bench.py times configs[1] only):
   1M bursts each, device-resident, fused demodulator, sliced soft bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from osmo_trx_amd import TrxHip, synth

n = int(os.environ.get("N_BURSTS", 1 << 20))
trx = TrxHip(0)


def run(name, iq, params, stride=148):
    dp = trx.params_tensor(params)
    res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
    soft = torch.empty((n, stride), dtype=torch.float32, device="cuda:0")
    f = lambda: trx.detect_demod(iq, dp, sps=4, soft_stride=stride, results=res, soft=soft)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    det = (trx.results_to_numpy(res)["rc"] > 0).mean()
    print(f"{name:44s} {ms:7.3f} ms  {n / ms / 1e3:7.1f} Mbursts/s  detected {det:.3f}")


iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4); run("configs[1] NB, max_toa 3", iq, p)
iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4, max_toa=63, delay_sym=(0.0, 60.0)); run("NB, max_toa 63", iq, p)
p2 = p.copy(); p2["max_toa"] = 112; run("NB, max_toa 112 (windowed path)", iq, p2)
iq, p, _ = synth.make_access_bursts(n, "cuda:0"); run("configs[2] RACH, max_toa 63", iq, p)
iq, p, _ = synth.make_access_bursts(n, "cuda:0", ext=True); run("configs[2] EXT_RACH (TS0/1/2), max_toa 63", iq, p)
iq, p = synth.make_mixed_bursts(n, "cuda:0"); run("configs[4] 7:1 NB:RACH mix (per GPU)", iq, p)
iq, p, _ = synth.make_edge_bursts(n, "cuda:0"); run("EDGE 8-PSK (444 soft bits)", iq, p, stride=444)

# 1 sample per symbol (configs[0] geometry, generic kernel)
iq1, p1, _ = synth.make_normal_bursts(n, "cuda:0", 1, burst_len=156)
dp1 = trx.params_tensor(p1)
res1 = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft1 = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
f1 = lambda: trx.detect_demod(iq1, dp1, sps=1, soft_stride=148, results=res1, soft=soft1)
f1(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): f1()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"{'configs[0] geometry: NB, 1 SPS, 156 samples':44s} {ms:7.3f} ms  {n / ms / 1e3:7.1f} Mbursts/s  detected {(trx.results_to_numpy(res1)['rc'] > 0).mean():.3f}")

# Viterbi alternative (cfg->use_va): its own kernel, one wave per burst
iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4)
x = torch.view_as_complex(iq.to(torch.float32).contiguous())
dp = trx.params_tensor(p)
f = lambda: trx.demod_va(x, dp)
f(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
print(f"{'Viterbi alternative (use_va), NB, demod only':44s} {ms:7.3f} ms  {n / ms / 1e3:7.1f} Mbursts/s")
