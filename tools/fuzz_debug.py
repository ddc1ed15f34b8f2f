import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib as O
from osmo_trx_amd import TrxHip
from test_gpu_parity import _fuzz_batch
trx = TrxHip(0)
L, ss = 625, 148
rng = np.random.default_rng(1000 * L + ss)
iq, params = _fuzz_batch(768, L, rng)
o_res, o_soft = O.pull_batch(iq.numpy(), 4, params, soft_stride=ss, slice_bits=True)
res, soft = trx.detect_demod(iq.to("cuda:0"), trx.params_tensor(params), sps=4, soft_stride=ss, exact=True)
g = trx.results_to_numpy(res)
big = params["max_toa"] > 112
live = big & ~np.isin(params["type"], [O.OFF, O.IDLE, O.SCH, 9]) & ~((params["tsc"] > 7) & np.isin(params["type"], [O.TSC, O.EDGE]))
o_res["rc"][live] = -3
bad = np.where(g["rc"] != o_res["rc"])[0]
print("n bad", len(bad))
for i in bad[:20]:
    print(i, params[i], "gpu", g[i], "orc", o_res[i], "absmax", int(np.abs(iq[i].numpy().astype(np.int32)).max()))
