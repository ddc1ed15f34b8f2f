#!/usr/bin/env python3
"""The fused kernels' FAST detector against the bit-exact kernel on the GPU itself (TRXHIP_FLAG_EXACT_DEMOD: the reference's
operand order everywhere; that kernel is pinned to the oracle by tests/test_gpu_parity.py), at full batch sizes:
rc / tsc / TOA identical?  max relative amp error, max C/I error against TRXHIP_FAST_CI_ATOL_DB, re-run rate.
   python tools/fast_detect_report.py [n_bursts] [workload ...]      workloads: nb nb63 rach ext mixed fuzz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from osmo_trx_amd import TrxHip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
loads = sys.argv[2:] or ["nb", "nb63", "rach", "ext", "mixed"]
trx = TrxHip(0)
dev = "cuda:0"
tot = {"bursts": 0, "reruns": 0}
for wl in loads:
    for rep in range(4 if wl == "nb" else 1):
        seed = synth.SEED + 7919 * rep
        if wl == "nb":
            iq, p, _ = synth.make_normal_bursts(n, dev, 4, seed=seed)
        elif wl == "nb63":
            iq, p, _ = synth.make_normal_bursts(n, dev, 4, seed=seed + 1, max_toa=33, delay_sym=(-2.0, 30.0))
        elif wl == "rach":
            iq, p, _ = synth.make_access_bursts(n, dev, seed=seed + 2)
        elif wl == "ext":
            iq, p, _ = synth.make_access_bursts(n, dev, seed=seed + 3, ext=True)
        elif wl == "mixed":
            iq, p = synth.make_mixed_bursts(n, dev, seed=seed + 4)
        else:
            raise SystemExit(wl)
        dp = trx.params_tensor(p)
        trx.fast_stats(reset=True)
        rf, sf = trx.detect_demod(iq, dp, sps=4, exact=False)
        st = trx.fast_stats(reset=True)
        re_, se = trx.detect_demod(iq, dp, sps=4, exact=True)
        torch.cuda.synchronize()
        f, e = trx.results_to_numpy(rf), trx.results_to_numpy(re_)
        same = all(np.array_equal(f[k], e[k]) for k in ("rc", "tsc", "clip", "idle", "nbits_div4")) and np.array_equal(f["toa"], e["toa"])
        det = e["rc"] > 0
        aref = np.hypot(e["amp_re"], e["amp_im"])[det]
        d = np.hypot(f["amp_re"] - e["amp_re"], f["amp_im"] - e["amp_im"])[det]
        okci = det & np.isfinite(e["ci"]) & np.isfinite(f["ci"])
        nan_same = np.array_equal(np.isnan(f["ci"]), np.isnan(e["ci"]))
        cie = np.abs(f["ci"] - e["ci"])[okci]
        bar = 1e-4 + 1.4e-5 * (1.0 + np.power(10.0, e["ci"][okci].astype(np.float64) * 0.1))
        tot["bursts"] += n
        tot["reruns"] += st["reruns"]
        print(f"{wl:6s} n={n} detected={int(det.sum())} rc/tsc/TOA identical={same}  reruns={st['reruns']} ({st['reruns'] / max(1, int(det.sum())):.4%} of detected)  "
              f"amp rel err max={float((d / aref).max()):.3e} p99.9={float(np.quantile(d / aref, 0.999)):.3e}  "
              f"ci err max={float(cie.max()):.3e} dB, max err/bar={float((cie / bar).max()):.3f} (NaN pattern identical={nan_same})  "
              f"soft max |fused - exact|={float((sf - se).abs().max()):.3e}", flush=True)
        del iq, rf, sf, re_, se
print("total", tot)
