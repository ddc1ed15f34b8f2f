#!/usr/bin/env python3
"""Per-phase cycle accounting of the normal-burst kernel (diagnostic build, -DTRX_DIAG): s_memtime deltas between the
   hand-placed blocks, per burst.   TRXHIP_LIB=.../libtrxhip_diag.so python tools/phase_cycles_nb.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth, trxhip
n = int(os.environ.get('N_BURSTS', str(1 << 20)))
trx = TrxHip(0)
L = trxhip.load_library()
L.trxhip_diag_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(params)
for _ in range(40):
    trx.detect_demod(iq, dp, sps=4)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
L.trxhip_diag_read(buf, 1)
trx.detect_demod(iq, dp, sps=4); torch.cuda.synchronize()
L.trxhip_diag_read(buf, 1)
names = {12: "loop top", 13: "wait for the prefetched samples", 14: "convert + LDS writes", 16: "flush previous burst's stores", 17: "take the work ticket (+ pool)",
         15: "issue next prefetch", 2: "DEC", 3: "CORR", 4: "AMAX", 5: "DETA", 6: "DETB", 10: "TAIL (record + demod)", 11: "slicer / loop end"}
tot = sum(buf[:20])
for k in (12, 13, 14, 16, 17, 15, 2, 3, 4, 5, 6, 10, 11):
    print(f"{names[k]:48s} {buf[k] / n:8.0f} cycles/burst  {100.0 * buf[k] / tot:5.1f} %")
print(f"{'total':48s} {tot / n:8.0f}")
import time
t0 = time.perf_counter()
for _ in range(20): trx.detect_demod(iq, dp, sps=4)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
