"""Sharded == unsharded on the HIP path, and bench.py's N > 1 code path executed.

A 1-GPU box is enough: two fresh processes, both on cuda:0, rendezvous over gloo (RCCL refuses two ranks on one device;
gloo carries the same device tensors through the host), each runs  broadcast_tables -> TrxHip(tables_blob=...) ->
detect_demod  on ITS shard_range() of one seeded 65536-burst mixed batch (normal + access bursts), generated shard-wise
with synth.make_mixed_bursts(offset=...).  The concatenation of the two ranks' result records and soft bits must equal
the single-process result bit for bit; max_over_ranks / sum_over_ranks / barrier run on device tensors.  Then bench.py
itself is launched the way the driver launches it for N = 2 (torch.distributed.run) with both ranks on cuda:0."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N = 65536
CHUNK = 4096            # 8 * CHUNK divides N / 2: both shards start on chunk boundaries of both generators

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from osmo_trx_amd import TrxHip, shard, synth
N, CHUNK, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rank, _, world = shard.init_distributed("gloo")
dev = "cuda:0"
torch.cuda.set_device(0)
blob = shard.broadcast_tables(dev)                                   # device tensors through the collective
trx = TrxHip(0, tables_blob=blob)
lo, hi = shard.shard_range(N, rank, world)
iq, params = synth.make_mixed_bursts(hi - lo, dev, seed=1234, chunk=CHUNK, offset=lo)
res, soft = trx.detect_demod(iq, trx.params_tensor(params), sps=4, soft_stride=148, slice_bits=True)
torch.cuda.synchronize()
shard.barrier()
tot = shard.sum_over_ranks(hi - lo, dev)
mx = shard.max_over_ranks(float(rank + 1), dev)
det = shard.sum_over_ranks(int((trx.results_to_numpy(res)["rc"] > 0).sum()), dev)
np.savez(out + f".{rank}.npz", res=res.cpu().numpy(), soft=soft.cpu().numpy(), lo=lo, hi=hi, tot=tot, mx=mx, det=det)
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_shards_on_the_gpu_equal_the_unsharded_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from osmo_trx_amd import TrxHip, synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(N), str(CHUNK), str(tmp_path / "shard")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
    parts = [np.load(str(tmp_path / f"shard.{r}.npz")) for r in range(2)]
    assert int(parts[0]["lo"]) == 0 and int(parts[0]["hi"]) == int(parts[1]["lo"]) == N // 2 and int(parts[1]["hi"]) == N
    assert float(parts[0]["tot"]) == float(parts[1]["tot"]) == N and float(parts[0]["mx"]) == float(parts[1]["mx"]) == 2.0
    # the same batch in one process, one launch
    trx = TrxHip(0)
    iq, params = synth.make_mixed_bursts(N, "cuda:0", seed=1234, chunk=CHUNK)
    res, soft = trx.detect_demod(iq, trx.params_tensor(params), sps=4, soft_stride=148, slice_bits=True)
    torch.cuda.synchronize()
    res, soft = res.cpu().numpy(), soft.cpu().numpy()
    assert np.array_equal(np.concatenate([parts[0]["res"], parts[1]["res"]]), res)          # 32-byte records, bit for bit
    assert np.array_equal(np.concatenate([parts[0]["soft"], parts[1]["soft"]]).view(np.uint32), soft.view(np.uint32))
    r = trx.results_to_numpy(torch.from_numpy(res))
    det = int((r["rc"] > 0).sum())
    assert float(parts[0]["det"]) == det and det > 0.8 * N
    assert ((r["rc"] == 3) | (r["rc"] == 1)).sum() == det                                   # normal and access bursts both found


def test_bench_py_two_ranks_code_path(tmp_path):
    """bench.py --gpus 2 launched as the driver launches it (torch.distributed.run, one process per rank), both ranks on
    cuda:0 over gloo: the N > 1 branch -- device-tensor broadcast of the tables, barriers, max-over-ranks timing, the
    weak-scaling value and the strong-scaling configs[4] leg on shard_range() -- runs end to end."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, TRXHIP_DIST_BACKEND="gloo", TRXHIP_ONE_DEVICE="1", TRXHIP_BENCH_STRONG_TOTAL=str(1 << 20))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--bursts", str(1 << 16), "--no-host-fed", "--legs", "strong", "--sustain-seconds", "0.2"]
    check_two_rank_line(subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT),
                        "torch.distributed.run / external")


def test_plain_bench_py_gpus_2_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment starts the two ranks itself (bench.spawn_ranks) and prints
    ONE line with n_gpus 2, the world, the backend and every rank's device."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TRXHIP_DIST_BACKEND="gloo", TRXHIP_ONE_DEVICE="1", TRXHIP_BENCH_STRONG_TOTAL=str(1 << 20))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--bursts", str(1 << 16), "--no-host-fed", "--legs", "strong", "--sustain-seconds", "0.2"]
    check_two_rank_line(subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT),
                        "bench.py self-spawn")


WORKER8 = r'''
import os, sys, time
T0 = time.time()
def stamp(what):
    print(f"[worker {os.environ.get('RANK')}] {what}: {time.time() - T0:.1f} s", file=sys.stderr, flush=True)
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from osmo_trx_amd import TrxHip, shard
N, base_path, out = int(sys.argv[2]), sys.argv[3], sys.argv[4]
rank, _, world = shard.init_distributed("gloo")
dev = "cuda:0"
torch.cuda.set_device(0)
blob = shard.broadcast_tables(dev)
trx = TrxHip(0, tables_blob=blob)
stamp("imports + rendezvous + tables + context")
lo, hi = shard.shard_range(N, rank, world)
base = np.load(base_path)
base_iq, base_p = torch.from_numpy(base["iq"]).to(dev), base["params"]
b = torch.arange(lo, hi, device=dev)
idx = ((b // 8 * 40503) % (len(base_p) // 8)) * 8 + b % 8          # burst b of the global batch (every 8th stays an access burst)
iq, params = base_iq[idx], base_p[idx.cpu().numpy()]
res, soft = trx.detect_demod(iq, trx.params_tensor(params), sps=4, soft_stride=148, slice_bits=True)
torch.cuda.synchronize()
stamp("shard tiled + detect_demod")
shard.barrier()
det = shard.sum_over_ranks(int((trx.results_to_numpy(res)["rc"] > 0).sum()), dev)
# position-weighted 64-bit checksums of the shard's records and soft bits (global row index: a row in the wrong place shows)
w = (b % 65521 + 1)[:, None]
r32 = res.view(torch.int32).to(torch.int64)
s32 = soft.view(torch.int32).to(torch.int64)
sums = [int(r32.sum()), int((r32 * w).sum()), int(s32.sum()), int((s32 * w).sum())]
sel = torch.from_numpy(np.random.default_rng(rank).choice(hi - lo, 2048, replace=False)).to(dev)
np.savez(out + f".{rank}.npz", lo=lo, hi=hi, det=det, sums=np.array(sums, dtype=np.int64), sel=sel.cpu().numpy(),
         res=res[sel].cpu().numpy(), soft=soft[sel].cpu().numpy())
stamp("checksums written")
dist.destroy_process_group()
'''


def test_eight_shards_of_the_8m_batch_equal_one_unsharded_launch(tmp_path):
    """BASELINE.json configs[4] in its real form, time-sliced on one GPU: a fixed 8M-burst mixed batch (7:1 NB:RACH) as 8
    contiguous shards -- 8 ranks over gloo, each  broadcast_tables -> TrxHip(tables_blob) -> detect_demod  on its 1M bursts --
    against ONE unsharded 8M-burst launch (21 GB of IQ, resident): position-weighted 64-bit checksums of every shard's result
    records and soft bits equal those of the same rows of the unsharded output, 2048 random rows per shard bit for bit.
    The batch is 65536 generated bursts tiled by a fixed index map (every rank reads the same base file: no generator, FFT
    or convolution library in the workers -- eight fresh processes compiling those at once cost 20 minutes)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 100 << 30:
        pytest.skip("needs ~60 GB of device memory")
    from osmo_trx_amd import TrxHip, synth
    n, world = 8 << 20, 8
    base_iq, base_p = synth.make_mixed_bursts(1 << 16, "cuda:0", seed=4321, chunk=4096)
    np.savez(str(tmp_path / "base.npz"), iq=base_iq.cpu().numpy(), params=base_p)
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    port = free_port()
    procs = []
    for r in range(world):                                        # (fresh child processes: nothing that touched the GPU is re-executed)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n), str(tmp_path / "base.npz"), str(tmp_path / "s8")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
    parts = [np.load(str(tmp_path / f"s8.{r}.npz")) for r in range(world)]
    assert [int(q["lo"]) for q in parts] == [r << 20 for r in range(world)] and int(parts[-1]["hi"]) == n
    trx = TrxHip(0)
    b = torch.arange(n, device="cuda:0")
    idx = ((b // 8 * 40503) % (len(base_p) // 8)) * 8 + b % 8
    iq, params = base_iq[idx], base_p[idx.cpu().numpy()]
    res, soft = trx.detect_demod(iq, trx.params_tensor(params), sps=4, soft_stride=148, slice_bits=True)
    torch.cuda.synchronize()
    del iq, idx
    r = trx.results_to_numpy(res)
    det = int((r["rc"] > 0).sum())
    assert det > 0.9 * n and all(float(q["det"]) == det for q in parts)
    assert ((r["rc"] == 3) | (r["rc"] == 1)).sum() == det and (r["rc"] == 3).sum() > 0.1 * n      # access and normal bursts both found
    for q in parts:
        lo, hi = int(q["lo"]), int(q["hi"])
        w = (b[lo:hi] % 65521 + 1)[:, None]
        r32 = res[lo:hi].view(torch.int32).to(torch.int64)
        s32 = soft[lo:hi].view(torch.int32).to(torch.int64)
        assert [int(r32.sum()), int((r32 * w).sum()), int(s32.sum()), int((s32 * w).sum())] == q["sums"].tolist(), lo
        sel = torch.from_numpy(q["sel"]).to("cuda:0") + lo
        assert np.array_equal(res[sel].cpu().numpy(), q["res"])
        assert np.array_equal(soft[sel].cpu().numpy().view(np.uint32), q["soft"].view(np.uint32))


def test_plain_bench_py_gpus_8_full_strong_batch(tmp_path):
    """`python bench.py --gpus 8`, self-spawned, every rank on cuda:0 over gloo, with configs[4]'s REAL fixed batch: 8388608
    mixed bursts, 1M per rank (the driver's 8-GPU run does the same over RCCL)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 100 << 30:
        pytest.skip("needs ~40 GB of device memory")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TRXHIP_BENCH_STRONG_TOTAL")}
    env.update(TRXHIP_DIST_BACKEND="gloo", TRXHIP_ONE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--bursts", str(1 << 16), "--no-host-fed", "--legs", "strong", "--sustain-seconds", "0"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 8 and j["world"] == 8 and j["backend"] == "gloo" and j["launcher"] == "bench.py self-spawn"
    assert len(j["devices"]) == 8 and all(d.startswith(f"rank {k}: cuda:0") for k, d in enumerate(j["devices"]))
    s = j["config"]["other_configs"]["configs[4]_strong"]
    assert s["scaling"] == "strong" and s["global_bursts"] == 8388608 and s["bursts_this_rank"] == 1 << 20
    assert 0.9 < s["detected_fraction"] < 1.0 and s["mbursts_per_s_all_gpus"] > 0


def check_two_rank_line(r, launcher):
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]                      # rank 0 alone prints
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["world"] == 2 and j["backend"] == "gloo" and j["launcher"] == launcher
    assert len(j["devices"]) == 2 and j["devices"][0].startswith("rank 0: cuda:0") and j["devices"][1].startswith("rank 1: cuda:0")
    # (the leg's launch count comes from the timed region's kernel time: two ranks sharing one GPU make that estimate rough)
    assert j["config"]["sustained"]["seconds"] >= 0.05 and j["config"]["sustained"]["mbursts_per_s_all_gpus"] > 0
    assert j["config"]["global_bursts"] == 2 << 16 and 0.9 < j["config"]["detected_fraction"] < 1.0
    assert j["value"] > 0 and j["roofline"]["kernel_ms"] > 0 and "cpu_baseline" not in j
    s = j["config"]["other_configs"]["configs[4]_strong"]
    assert s["scaling"] == "strong" and s["global_bursts"] == 1 << 20 and s["bursts_this_rank"] == 1 << 19
    assert 0.85 < s["detected_fraction"] < 1.0


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from osmo_trx_amd import TrxHip, shard, trxhip
rank, local_rank, world = shard.init_distributed()                    # WORLD_SIZE=1 TRXHIP_DIST_BACKEND=nccl: one rank over RCCL
assert shard.backend_name() == "nccl" and world == 1
dev = "cuda:0"
blob = shard.broadcast_tables(dev)                                    # uint8 + int64 broadcasts on device tensors, checksum verified
assert blob == trxhip.generate_tables_host()
assert shard.max_over_ranks(1.25, dev) == 1.25 and shard.sum_over_ranks(3.0, dev) == 3.0
assert shard.gather_strings("cuda:0 test", dev) == ["cuda:0 test"]
shard.barrier()
trx = TrxHip(0, tables_blob=blob)                                     # the broadcast blob is what the context adopts
torch.cuda.synchronize()
import torch.distributed as dist
dist.destroy_process_group()
print("rccl first contact ok", len(blob))
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_first_contact_world_1():
    """RCCL (torch.distributed "nccl") exercised with one rank in a FRESH process -- nothing in it has touched the GPU before the
    process group is created: the table broadcast of shard.broadcast_tables on device tensors, max / sum over ranks, the
    all_gather of gather_strings, barrier.  The multi-rank logic is the gloo tests'; this is the backend."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TRXHIP_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", RCCL_WORKER, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rccl first contact ok" in r.stdout


def test_bench_main_leg_over_rccl_world_1():
    """bench.py --gpus 1 with WORLD_SIZE=1 TRXHIP_DIST_BACKEND=nccl: the line says backend nccl, and its collectives (table
    broadcast, max over ranks of the timings, device list) ran on RCCL"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TRXHIP_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--main-only", "--steps", "3", "--warmup", "1",
                        "--bursts", "65536"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["backend"] == "nccl" and d["n_gpus"] == 1 and d["value"] > 0
