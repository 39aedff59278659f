"""world_size-2 gloo tests (CPU) of the N>1 plumbing: read sharding and the variable-size table
exchange that Engine.allreduce_table runs over RCCL on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vfind_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfind_amd.dist import allgather_blobs
        # blobs of different sizes and contents per rank
        n = 1000 + 777 * rank
        blob = (torch.arange(n, dtype=torch.int64) * (rank + 3) % 251).to(torch.uint8)
        parts = allgather_blobs(blob, n)
        ok = len(parts) == world
        for r, (buf, nbytes) in enumerate(parts):
            m = 1000 + 777 * r
            want = (torch.arange(m, dtype=torch.int64) * (r + 3) % 251).to(torch.uint8)
            ok = ok and nbytes == m and torch.equal(buf[:m], want)
        # a send buffer already padded beyond its valid size is handled too
        padded = torch.zeros(4096, dtype=torch.uint8)
        padded[:n] = blob
        parts2 = allgather_blobs(padded, n)
        ok = ok and all(torch.equal(a[0][:a[1]], b[0][:b[1]]) for a, b in zip(parts, parts2))
        # weak-scaling bookkeeping: the per-rank counts all-reduce to the global total
        lo, hi = shard_range(10_000_019, rank, world)
        t = torch.tensor([hi - lo], dtype=torch.int64)
        dist.all_reduce(t)
        ok = ok and int(t.item()) == 10_000_019
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allgather_blobs_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("n,world", [(10, 3), (0, 4), (7, 8), (1_000_000_007, 8)])
def test_shard_range_partitions(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks (before touching any GPU) and report
    n_gpus == 2; a mismatch between --gpus and an existing WORLD_SIZE is refused"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--dry-run",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["max_over_ranks"] == 2.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
