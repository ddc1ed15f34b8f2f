#!/usr/bin/env python3
"""One line per workload for the library TRXHIP_LIB points at (default: the product): the fused demodulator's soft bits against
the bit-exact kernel's on the same GPU (which tests/ pin to the oracle), and its rate.  Used to price the composite filter's tap
window (TRX_FUSED_U0 / TRX_FUSED_NT measurement builds, tools/build_variants.py t28:all:-DTRX_FUSED_U0=4,-DTRX_FUSED_NT=28):
   for L in ...; do TRXHIP_LIB=$PWD/$L python tools/fused_taps_report.py; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from osmo_trx_amd import TrxHip, synth

trx = TrxHip(0)
dev = "cuda:0"
name = os.path.basename(os.environ.get("TRXHIP_LIB", "libtrxhip.so"))
for wl, n in (("normal", 1 << 20), ("access", 1 << 18)):
    if wl == "normal":
        iq, p, _ = synth.make_normal_bursts(n, dev, 4, seed=0xB17E)
    else:
        iq, p, _ = synth.make_access_bursts(n, dev, seed=0xB17F)
    dp = trx.params_tensor(p)
    re_, se = trx.detect_demod(iq, dp, sps=4, exact=True)
    rf, sf = trx.detect_demod(iq, dp, sps=4, exact=False)
    torch.cuda.synchronize()
    e = trx.results_to_numpy(re_)
    det = torch.from_numpy(e["rc"] > 0).to(dev)
    amp = torch.from_numpy((e["amp_re"].astype("f8") ** 2 + e["amp_im"].astype("f8") ** 2) ** 0.5).to(dev)
    rms = torch.from_numpy(e["energy"].astype("f8").clip(0) ** 0.5).to(dev)
    scale = torch.clamp(rms / (4.0 * amp.clamp_min(1e-30)), min=1.0)[det][:, None]
    err = (sf[det].double() - se[det].double()).abs()
    real = (scale[:, 0] <= 1.0)
    raw_e, raw_f = 2.0 * se[det].double() - 1.0, 2.0 * sf[det].double() - 1.0
    rel = (raw_f - raw_e).abs() / raw_e.abs().clamp_min(1e-30)
    r05 = float(rel[raw_e.abs() >= 0.05].max())
    r25 = float(rel[raw_e.abs() >= 0.25].max())
    be, bf = torch.round(255.0 * se[det]), torch.round(255.0 * sf[det])
    nbyte = int((be != bf).sum())
    # rate of the fused kernel
    for _ in range(3):
        trx.detect_demod(iq, dp, sps=4, results=rf, soft=sf)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); trx.detect_demod(iq, dp, sps=4, results=rf, soft=sf); b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print(f"{name:22s} {wl:7s} n={n} detected={int(det.sum())}  max|soft-ref| {float(err.max()):.3e} (sliced 0..1), on real detections "
          f"(rms <= 4|amp|) {float(err[real].max()):.3e}, over the bar's scale {float((err / scale).max()):.3e};  max relative error of the raw "
          f"soft value: {r05:.3e} (|soft| >= 0.05) {r25:.3e} (>= 0.25);  TRXD soft bytes differing {nbyte} of {err.numel()} ({nbyte / err.numel():.2e});  "
          f"{n / ms / 1e3:.1f} Mbursts/s ({ms:.4f} ms)", flush=True)
    del iq, re_, se, rf, sf
