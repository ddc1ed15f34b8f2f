#!/usr/bin/env python3
"""RCCL smoke on one GPU (world size 1): the collectives bench.py uses at N > 1 (uint8 / int64 broadcast of the table
blob, float64 all-reduce MAX) through torch.distributed's "nccl" backend.  Checks that the backend loads and runs on
this image; the multi-rank logic itself is covered by the gloo tests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
import torch
import torch.distributed as dist
from osmo_trx_amd import trxhip

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
blob = trxhip.generate_tables_host()
t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to("cuda:0")
ck = torch.tensor([trxhip.tables_checksum(blob) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device="cuda:0")
dist.broadcast(t, src=0); dist.broadcast(ck, src=0)
v = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
dist.all_reduce(v, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert t.cpu().numpy().tobytes() == blob and float(v.item()) == 1.25
print("rccl ok:", len(blob), "byte table blob broadcast, all-reduce MAX, barrier")
dist.destroy_process_group()
