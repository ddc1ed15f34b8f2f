"""world_size-2 gloo tests (CPU): the N>1 path of bench.py -- contiguous sharding with no data-path
collective, and the init-time table broadcast with checksum verification."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from osmo_trx_amd import shard, trxhip
rank, local_rank, world = shard.init_distributed("gloo")
assert world == int(os.environ["WORLD_SIZE"]) and dist.get_backend() == "gloo"
blob = shard.broadcast_tables()                      # rank 0 generates, rank 1 receives + verifies checksum
ck = trxhip.tables_checksum(blob)
lo, hi = shard.shard_range(1000003, rank, world)
tot = shard.sum_over_ranks(hi - lo)
mx = shard.max_over_ranks(float(rank + 1))
names = shard.gather_strings(f"rank {rank}: cpu worker")          # bench.py's device list (same call on RCCL)
shard.barrier()
print(json.dumps({"rank": rank, "ck": ck, "lo": lo, "hi": hi, "tot": tot, "mx": mx, "n": len(blob), "names": names,
                  "backend": shard.backend_name()}))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_table_broadcast_and_sharding(tmp_path):
    from osmo_trx_amd import build as trx_build, trxhip
    trx_build.build_lib()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        import json
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    local = trxhip.generate_tables_host()
    assert outs[0]["ck"] == outs[1]["ck"] == trxhip.tables_checksum(local)
    assert outs[0]["n"] == len(local)
    assert outs[0]["lo"] == 0 and outs[0]["hi"] == outs[1]["lo"] and outs[1]["hi"] == 1000003
    assert outs[0]["tot"] == outs[1]["tot"] == 1000003
    assert outs[0]["mx"] == outs[1]["mx"] == 2.0
    assert outs[0]["names"] == outs[1]["names"] == ["rank 0: cpu worker", "rank 1: cpu worker"]
    assert outs[0]["backend"] == "gloo"


def test_shard_ranges_partition_exactly():
    from osmo_trx_amd import shard
    for n in (0, 1, 7, 8, 1 << 20, 8388608, 1000003):
        for world in (1, 2, 4, 8):
            r = [shard.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_aligned_shard_ranges_for_every_world_size():
    """bench.py's strong-scaling leg: shard edges on multiples of the generator's block (8 * chunk) for ANY N -- round 3's
    plain shard_range() crashed the leg for N = 3, 5, 6, 7 (offsets that are no multiple of 524288)."""
    from osmo_trx_amd import shard
    for n, align in ((8 << 20, 8 * 65536), (1 << 20, 8 * 4096), (1000003, 4096)):
        for world in range(1, 17):
            r = [shard.shard_range_aligned(n, k, world, align) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert all(a % align == 0 for a, _ in r)
            assert all(b > a for a, b in r) or n < world * align
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= align + (n % align)


def test_bench_self_spawn_refuses_more_ranks_than_gpus():
    """plain `python bench.py --gpus 2` launches its own ranks; on a box without that many GPUs the rank without a device says
    so and takes the others down, instead of the job running as world 1 and printing n_gpus 1 (round 3).  (Round 5: the
    launcher itself no longer counts devices -- torch.cuda.device_count() may fall back to a HIP call there)"""
    import pytest
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node: the launch would succeed (tests/test_gpu_sharded.py covers it)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRXHIP_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    assert "but this node shows" in r.stderr or "bench.py needs a GPU" in r.stderr


def test_eight_rank_world_matches_the_eight_gpu_layout(tmp_path):
    """BASELINE.json configs[4]'s layout -- 8 ranks -- over gloo on the CPU: the table broadcast reaches every rank with the
    same checksum, the contiguous shards of 1,000,003 bursts partition exactly, max / sum reductions and the device-list
    gather return the same on every rank.  (The 8-GPU run itself needs an 8-GPU node; this is its control path.)"""
    import json
    from osmo_trx_amd import build as trx_build, trxhip
    trx_build.build_lib()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    world = 8
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    ck = trxhip.tables_checksum(trxhip.generate_tables_host())
    assert all(o["ck"] == ck for o in outs)
    assert outs[0]["lo"] == 0 and outs[-1]["hi"] == 1000003 and all(outs[k]["hi"] == outs[k + 1]["lo"] for k in range(world - 1))
    assert all(o["tot"] == 1000003 and o["mx"] == float(world) for o in outs)
    assert all(o["names"] == [f"rank {k}: cpu worker" for k in range(world)] for o in outs)
