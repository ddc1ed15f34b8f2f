#!/usr/bin/env python3
"""The Viterbi alternative alone (for rocprofv3 --pmc passes): 1M normal bursts, a warm-up and REPS launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from osmo_trx_amd import TrxHip, synth
n = int(os.environ.get("N_BURSTS", 1 << 20))
trx = TrxHip(0)
iq, p, _ = synth.make_normal_bursts(n, "cuda:0", 4)
x = torch.view_as_complex(iq.to(torch.float32).contiguous())
dp = trx.params_tensor(p)
for _ in range(1 + int(os.environ.get("REPS", 2))):
    trx.demod_va(x, dp)
torch.cuda.synchronize()
