#!/usr/bin/env python3
"""Per-phase cycle accounting of the fused kernel (diagnostic build): s_memtime deltas per phase, per burst.
   TRXHIP_LIB=.../libtrxhip_diag.so [TRXHIP_WPB=1] [WORKLOAD=normal|rach|ext|exact] python tools/phase_cycles.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth, trxhip
n = int(os.environ.get('N_BURSTS', str(1 << 17)))
trx = TrxHip(0)
L = trxhip.load_library()
L.trxhip_diag_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
wl = os.environ.get('WORKLOAD', 'normal')
if wl in ('rach', 'ext'):
    iq, params, _ = synth.make_access_bursts(n, "cuda:0", ext=(wl == 'ext'))
else:
    iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
kw = dict(soft_stride=148, slice_bits=True, exact=(wl == 'exact'))
dp = trx.params_tensor(params)
trx.detect_demod(iq, dp, _diag_mask=int(os.environ.get('DIAG_MASK', '0'), 0), **kw); torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
L.trxhip_diag_read(buf, 1)
trx.detect_demod(iq, dp, _diag_mask=int(os.environ.get('DIAG_MASK', '0'), 0), **kw); torch.cuda.synchronize()
L.trxhip_diag_read(buf, 1)
names = ["0 load/convert", "1 clip/energy/rssi", "2 decimate", "3 correlate", "4 argmax+gate", "5 peak ratio", "6 bisection",
         "7 C/I + amp", "8 demod setup", "9 edge round(s)", "10 composite FIR", "11 epilogue+stores", "12 result record"]
names += ["13 wait first prefetched dword", "14 convert + LDS writes", "15 issue next prefetch", "16 flush previous burst's output", "17 take the work ticket"]
tot = sum(buf[:18])
for i, nm in enumerate(names):
    print(f"{nm:22s} {buf[i] / n:8.0f} cycles/burst  {100.0 * buf[i] / tot:5.1f} %")
print(f"{'total':22s} {tot / n:8.0f}")
import time
t0 = time.perf_counter()
for _ in range(5): trx.detect_demod(iq, dp, _diag_mask=int(os.environ.get('DIAG_MASK', '0'), 0), **kw)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms  {n * 5 / (time.perf_counter() - t0) / 1e6:.1f} Mbursts/s")
