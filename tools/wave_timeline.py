#!/usr/bin/env python3
"""Start / end wall-clock of every wave of the fused kernel (diagnostic build): is the persistent grid balanced?
   TRXHIP_LIB=.../libtrxhip_diag.so [WORKLOAD=normal|mixed|rach] python tools/wave_timeline.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from osmo_trx_amd import TrxHip, synth, trxhip
n = int(os.environ.get('N_BURSTS', str(1 << 20)))
trx = TrxHip(0)
L = trxhip.load_library()
L.trxhip_diag_read_waves.argtypes = [C.c_void_p, C.c_int]
wl = os.environ.get('WORKLOAD', 'normal')
if wl == 'mixed':
    iq, params = synth.make_mixed_bursts(n, "cuda:0")
elif wl == 'rach':
    iq, params, _ = synth.make_access_bursts(n, "cuda:0")
else:
    iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
dp = trx.params_tensor(params)
for _ in range(3):
    trx.detect_demod(iq, dp, soft_stride=148, slice_bits=True); torch.cuda.synchronize()
W = 4096
buf = np.zeros((W, 24), dtype=np.uint64)
L.trxhip_diag_read_waves(buf.ctypes.data, W)
t0, t1 = buf[:, 20].astype(np.int64), buf[:, 21].astype(np.int64)
hw, xcc = buf[:, 22].astype(np.int64), buf[:, 23].astype(np.int64) & 0xf
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds (100 MHz)
print(f"kernel span {e.max():.1f} us; wave start min/median/max {s.min():.1f}/{np.median(s):.1f}/{s.max():.1f} us; "
      f"wave end min/median/max {e.min():.1f}/{np.median(e):.1f}/{e.max():.1f} us")
print(f"mean residency {(e - s).mean() / e.max():.3f} of the span; busy (end-start) min/median/max {(e - s).min():.1f}/{np.median(e - s):.1f}/{(e - s).max():.1f} us")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"XCC {x}: {m.sum():4d} waves, start {s[m].min():7.1f}..{s[m].max():7.1f}, end {e[m].min():7.1f}..{e[m].max():7.1f}, mean busy {(e - s)[m].mean():7.1f} us")
blk = np.arange(W) // 16
eb = np.array([e[blk == b].max() for b in range(W // 16)])
sb = np.array([s[blk == b].min() for b in range(W // 16)])
print("per workgroup: end percentiles 0/10/50/90/100:", np.percentile(eb, [0, 10, 50, 90, 100]).round(1), " start:", np.percentile(sb, [0, 50, 100]).round(1))
print("within a workgroup: spread of wave ends (max-min) median/max:", np.median([np.ptp(e[blk == b]) for b in range(W // 16)]).round(1), max(np.ptp(e[blk == b]) for b in range(W // 16)).round(1))
