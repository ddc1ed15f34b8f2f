"""Multi-GPU plumbing: one process per GPU, bursts sharded by contiguous ranges, tables broadcast once.

Bursts are independent: rank g of G owns bursts [g*N/G, (g+1)*N/G) and there is NO
data-path collective.  The only collective is the init-time broadcast of the ~45 KB table blob from rank 0
(RCCL over xGMI when the backend is "nccl", gloo in the CPU tests), verified by checksum on every rank.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import trxhip


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None):
    rank, local_rank, world = env_world()
    # world 1 with WORLD_SIZE=1 and TRXHIP_DIST_BACKEND spelled out: the process group is still created, so that the collectives of
    # the N > 1 path (table broadcast, max / sum over ranks, barrier) run on the named backend with one rank -- RCCL's first
    # contact on a 1-GPU box (tests/test_gpu_sharded.py)
    forced = world == 1 and "WORLD_SIZE" in os.environ and bool(os.environ.get("TRXHIP_DIST_BACKEND"))
    if (world > 1 or forced) and not dist.is_initialized():
        # before anything initialises the GPU runtime: dmabuf IPC (RCCL peer buffers on hosts without the legacy mode)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # TRXHIP_DIST_BACKEND=gloo: ranks that share one GPU (tests/test_gpu_sharded.py runs bench.py's N = 2 path on
            # a 1-GPU box; RCCL refuses two ranks on one device, gloo moves the same device tensors through the host)
            backend = os.environ.get("TRXHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_total, rank, world):
    """Contiguous range of bursts owned by `rank`: [lo, hi)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def shard_range_aligned(n_total, rank, world, align):
    """shard_range() with every inner boundary rounded down to a multiple of `align` (the last shard ends at n_total):
    for generators and layouts that can only start on a block boundary (synth.make_mixed_bursts: 8 * chunk).  Any world size
    works -- N = 3, 5, 6, 7 included; shards differ by at most `align` bursts."""
    def edge(r):
        return n_total if r >= world else ((n_total * r) // world) // align * align
    return edge(rank), edge(rank + 1)


def backend_name():
    return dist.get_backend() if dist.is_initialized() else None


def gather_strings(text, device=None, width=96):
    """Every rank's `text` (ASCII, cut to `width` bytes) on every rank, in rank order: one all_gather of fixed-size byte
    tensors -- works on RCCL (device tensors) and gloo alike."""
    rank, _, world = env_world()
    if not dist.is_initialized():
        return [text]
    dev = torch.device(device) if device is not None else torch.device("cpu")
    raw = text.encode("ascii", "replace")[:width].ljust(width, b"\0")
    mine = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    parts = [torch.zeros(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [bytes(t.cpu().numpy().tobytes()).rstrip(b"\0").decode("ascii", "replace") for t in parts]


def broadcast_tables(device=None):
    """Rank 0 generates the table blob on the host; everyone receives it (RCCL/gloo broadcast) and checks
    the FNV-1a checksum.  Returns the blob as bytes."""
    rank, _, world = env_world()
    size = int(trxhip.load_library().trxhip_tables_size())
    if not dist.is_initialized():
        return trxhip.generate_tables_host()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if rank == 0:
        blob = trxhip.generate_tables_host()
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        ck = torch.tensor([trxhip.tables_checksum(blob) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev)
    else:
        t = torch.zeros(size, dtype=torch.uint8, device=dev)
        ck = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    dist.broadcast(ck, src=0)
    blob = t.cpu().numpy().tobytes()
    if (trxhip.tables_checksum(blob) & 0x7FFFFFFFFFFFFFFF) != int(ck.item()):
        raise trxhip.TrxHipError(f"rank {rank}: table blob checksum mismatch after broadcast")
    return blob


def max_over_ranks(value, device=None):
    rank, _, world = env_world()
    if not dist.is_initialized():
        return float(value)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    rank, _, world = env_world()
    if not dist.is_initialized():
        return float(value)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
