"""Which records differ between the split path and the general kernel alone (diagnostic for tests/test_gpu_nb_kernel.py)."""
import numpy as np
import torch
from osmo_trx_amd import TrxHip, synth

n = 65536
mixed = synth.make_mixed_bursts(n, "cpu")
rach = synth.make_access_bursts(n, "cpu", seed=91)[:2]
nb = synth.make_normal_bursts(n, "cpu", 4, seed=92)[:2]
t2 = TrxHip(0)
for name, (iq, params) in (("mixed", mixed), ("rach", rach), ("nb", nb), ("mixed2", mixed)):
    d_iq, d_p = iq.to("cuda:0"), t2.params_tensor(params)
    t2.set_nb_kernel(False)
    ref_res, ref_soft = t2.detect_demod(d_iq, d_p, sps=4)
    torch.cuda.synchronize()
    t2.set_nb_kernel(True)
    for k in range(6):
        res, soft = t2.detect_demod(d_iq, d_p, sps=4)
        torch.cuda.synchronize()
        a, b = t2.results_to_numpy(res), t2.results_to_numpy(ref_res)
        bad = np.zeros(n, bool)
        for f in a.dtype.names:
            same = (a[f] == b[f]) | ((a[f] != a[f]) & (b[f] != b[f])) if a[f].dtype.kind == "f" else (a[f] == b[f])
            if not same.all():
                idx = np.flatnonzero(~same)
                print(name, k, f, len(idx), idx[:6], a[f][idx[:6]], b[f][idx[:6]], "type", params["type"][idx[:6]], "tsc", params["tsc"][idx[:6]])
            bad |= ~same
        sb = (soft != ref_soft).any(dim=1).cpu().numpy()
        print(name, k, "records differing", int(bad.sum()), "soft rows differing", int(sb.sum()), t2.fast_stats())
t2.close()
