#!/usr/bin/env python3
"""Host pipe, PCIe-inclusive: staged submit (the DMA engine uploads the pinned slot) against submit_by_ref (a device kernel
fetches the bursts from a registered host ring through their addresses).  Same batches, all slots in flight.
   python tools/bench_byref.py [bursts_per_submit] [slots] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from osmo_trx_amd import TrxHip, synth
from osmo_trx_amd.trxhip import HostPipe, TRXD_META_DTYPE

hb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
trx = TrxHip(0)
iq, params = synth.make_mixed_bursts(hb * depth, "cpu")
ring = np.ascontiguousarray(iq.numpy())                      # the "receive ring": hb * depth bursts
pipe = HostPipe(trx, hb, depth=depth, soft_stride=0, pkt_stride=160)
pipe.register_host(ring)
for sl in range(depth):
    v = pipe.slot(sl)
    v["iq"][:] = ring[sl * hb:(sl + 1) * hb]
    v["params"][:] = params[sl * hb:(sl + 1) * hb]
    v["meta"][:] = np.zeros(hb, dtype=TRXD_META_DTYPE)
    v["meta"]["version"] = 1
    pipe.sources(sl)[:] = ring.ctypes.data + 2500 * (sl * hb + np.arange(hb, dtype=np.uint64))
for name, submit in (("staged", pipe.submit), ("by_ref", pipe.submit_by_ref), ("staged", pipe.submit), ("by_ref", pipe.submit_by_ref)):
    for sl in range(depth):
        submit(sl, hb)
    for sl in range(depth):
        pipe.wait(sl)
    t0 = time.perf_counter()
    for it in range(iters):
        sl = it % depth
        pipe.wait(sl)
        submit(sl, hb)
    for sl in range(depth):
        pipe.wait(sl)
    t = time.perf_counter() - t0
    print(f"{name}: {iters * hb / t / 1e6:.2f} Mbursts/s  ({iters * hb * 2500 / t / 1e9:.1f} GB/s of samples over the link), {hb} bursts x {depth} slots")
pipe.close()
