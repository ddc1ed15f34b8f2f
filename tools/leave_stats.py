import sys, torch
sys.path.insert(0, '/root/repo')
from osmo_trx_amd import TrxHip, synth
trx = TrxHip(0)
n = 1 << 20
iq, params, _ = synth.make_normal_bursts(n, "cuda:0", 4)
d_p = trx.params_tensor(params)
trx.fast_stats(reset=True)
res, soft = trx.detect_demod(iq, d_p, sps=4)
torch.cuda.synchronize()
print("configs[1] 1M:", trx.fast_stats(reset=True))
