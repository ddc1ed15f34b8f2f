import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/osmo_trx_amd") else os.getcwd())
import torch, numpy as np
from osmo_trx_amd import TrxHip, synth
n = 1 << 20
trx = TrxHip(0)
res = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
soft = torch.empty((n, 148), dtype=torch.float32, device="cuda:0")
def t(iq, p, reps=20):
    dp = trx.params_tensor(p)
    f = lambda: trx.detect_demod(iq, dp, sps=4, soft_stride=148, slice_bits=True, results=res, soft=soft)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
iq_n, p_n, _ = synth.make_normal_bursts(n, "cuda:0", 4)
iq_r, p_r, _ = synth.make_access_bursts(n, "cuda:0")
iq_m, p_m = synth.make_mixed_bursts(n, "cuda:0")
# a mix with the access bursts in ONE block at the end instead of every 8th burst (same counts)
idx = np.concatenate([np.arange(n)[np.arange(n) % 8 != 7], np.arange(n)[np.arange(n) % 8 == 7]])
iq_b = iq_m[torch.from_numpy(idx).to("cuda:0")].contiguous(); p_b = p_m[idx]
for rnd in range(2):
    tn, tr, tm, tb = t(iq_n, p_n), t(iq_r, p_r), t(iq_m, p_m), t(iq_b, p_b)
    print(f"normal {tn:.4f} ms  rach {tr:.4f}  mixed {tm:.4f}  (7/8 n + 1/8 r = {(7*tn+tr)/8:.4f}; ratio {tm/((7*tn+tr)/8):.3f})  blocked mix {tb:.4f}")
