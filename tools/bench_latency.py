#!/usr/bin/env python3
"""Small-batch latency of trxhip_detect_demod_batch (what a transceiver that batches one frame of a few ARFCNs sees):
time per call for n = 1 .. 4096 device-resident bursts, launched directly and replayed from a captured HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth

trx = TrxHip(0)
for n in (1, 8, 64, 512, 4096):
    iq, params, _ = synth.make_normal_bursts(max(n, 64), "cuda:0", 4)
    iq, params = iq[:n].contiguous(), params[:n]
    dp = trx.params_tensor(params)
    res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
    soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
    f = lambda: trx.detect_demod(iq, dp, results=res, soft=soft)
    for _ in range(20): f()
    torch.cuda.synchronize()
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / reps * 1e6
    ref = res.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            f()
    res.zero_()
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(res, ref))
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / reps * 1e6
    print(f"n = {n:5d}: direct {direct:7.1f} us/call ({n / direct:7.2f} Mbursts/s)   graph replay {graph:7.1f} us/call ({n / graph:7.2f} Mbursts/s)   identical {same}")
