#!/usr/bin/env python3
"""Round trip of one host-fed batch through trxhip_hostpipe_* (submit -> wait, nothing else in flight), the number the
gather stage's throughput under the reference's 32-deep FIFO rule hangs on:
   python tools/bench_hostpipe_rt.py [n_bursts ...]     -> microseconds per round trip, soft rows / TRXD datagrams / both"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from osmo_trx_amd import TrxHip, synth
from osmo_trx_amd.trxhip import HostPipe

sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]
trx = TrxHip(0)
iq, params, _ = synth.make_normal_bursts(max(sizes), "cpu", 4)
iq = iq.numpy()
for n in sizes:
    for name, kw in (("soft", dict(soft_stride=148, pkt_stride=0)), ("trxd", dict(soft_stride=0, pkt_stride=160)),
                     ("soft+trxd", dict(soft_stride=148, pkt_stride=160))):
        pipe = HostPipe(trx, n, depth=2, **kw)
        s = pipe.slot(0)
        s["iq"][:n] = iq[:n]
        s["params"][:n] = params[:n]
        if s["meta"] is not None:
            s["meta"]["version"] = 1
        for _ in range(20):
            pipe.submit(0, n); pipe.wait(0)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.submit(0, n); pipe.wait(0)
        dt = (time.perf_counter() - t0) / reps
        det = int((s["results"]["rc"][:n] > 0).sum())
        print(f"n {n:5d} {name:10s} round trip {dt * 1e6:8.1f} us   {n / dt / 1e6:7.2f} Mbursts/s serial   detected {det}")
        pipe.close()
