#!/usr/bin/env python3
"""Throughput of the multi-ARFCN front-end kernels (BASELINE.json configs[3]): Channelizer(4,192,16)::rotate over
256k blocks, Resampler(65,48) on the channel streams, then burst detection on the 4-SPS channel streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth

n_blocks = int(os.environ.get("NBLOCKS", 1 << 18))
trx = TrxHip(0)
wide = synth.make_wideband_stream(n_blocks, "cuda:0")


def timeit(fn, reps=20):
    keep = [fn(), fn()]; torch.cuda.synchronize()      # two live outputs: the timed calls below allocate nothing new
    del keep
    for _ in range(60):                                # settled clocks (the GPU needs tens of ms of load)
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


ms, ch = timeit(lambda: trx.channelize(wide, n_blocks))
byt = n_blocks * (768 * 4 + 4 * 192 * 8)
print(f"channelize: {ms:.3f} ms for {n_blocks} blocks = {n_blocks / ms / 1e3:.1f} Mblocks/s, {byt / ms / 1e6:.0f} GB/s ({byt / ms / 1e6 / 8000:.1%} of 8 TB/s)")
n_in = (ch.shape[1] // 48) * 48
x = ch[:, :n_in].contiguous()
ms, rs = timeit(lambda: trx.resample(x, 65, 48))
byt = 4 * n_in * 8 + 4 * (n_in // 48 * 65) * 8
print(f"resample 65/48: {ms:.3f} ms, {byt / ms / 1e6:.0f} GB/s ({byt / ms / 1e6 / 8000:.1%} of 8 TB/s)")

# the fused front end (trxhip_rx_frontend_pull: channelizer + resampler in one pass)
from osmo_trx_amd import trxhip
fe = trxhip.RxFrontEnd(trx)
ms, out = timeit(lambda: fe.pull(wide, n_blocks))
n_out = n_blocks * 192 // 48 * 65
byt = n_blocks * 768 * 4 + 4 * n_out * 8
print(f"rx_frontend_pull (fused): {ms:.3f} ms = {n_blocks / ms / 1e3:.1f} Mblocks/s, {byt / ms / 1e6:.0f} GB/s ({byt / ms / 1e6 / 8000:.1%} of 8 TB/s)")
fe.close()
