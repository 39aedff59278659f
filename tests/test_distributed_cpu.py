"""world_size-2 and -3 gloo tests (CPU) of the N>1 plumbing: read sharding, the launcher, and the owner-partitioned table
exchange of vfind_amd.dist.exchange_shares (Engine.allreduce_table's torch transport) over a dict-backed stand-in engine."""
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


class _DictEngine:
    """stand-in for vfind_amd.Engine on CPU tensors: a {key: count} dict behind the raw-address methods exchange_shares uses.
    Blob = 16-byte header {entries, key bytes} + per entry {u32 key length, u64 count} + the keys; owner of a key = crc32 mod world
    (the product's owner function lives in the kernels; the exchange only needs every rank to agree on one)"""
    def __init__(self, table):
        import torch
        self.table = dict(table)
        self.torch_device = torch.device("cpu")
        self.device = None
        self.calls = []

    @staticmethod
    def _owner(key, world):
        import zlib
        return zlib.crc32(key) % world

    @staticmethod
    def _blob(items):
        import struct
        items = sorted(items)
        body = b"".join(struct.pack("<IQ", len(k), c) for k, c in items) + b"".join(k for k, _ in items)
        return struct.pack("<QQ", len(items), sum(len(k) for k, _ in items)) + body

    @staticmethod
    def _parse(raw):
        import struct
        n, kb = struct.unpack_from("<QQ", raw, 0)
        at, lens = 16, []
        for _ in range(n):
            ln, c = struct.unpack_from("<IQ", raw, at)
            lens.append((ln, c))
            at += 12
        out = []
        for ln, c in lens:
            out.append((raw[at:at + ln], c))
            at += ln
        assert at == len(raw) and sum(l for l, _ in lens) == kb
        return out

    def _share(self, world, owner):
        return self._blob([(k, c) for k, c in self.table.items() if self._owner(k, world) == owner])

    def partition_sizes(self, world):
        return [len(self._share(world, o)) for o in range(world)]

    def partition_export(self, world, owner, dst_ptr, capacity):
        import ctypes
        b = self._share(world, owner)
        assert len(b) <= capacity, "exchange_shares handed a bucket too little room"
        ctypes.memmove(dst_ptr, b, len(b))
        return len(b)

    def clear(self):
        self.table = {}

    def merge_from(self, src_ptr, nbytes):
        import ctypes
        for k, c in self._parse(ctypes.string_at(src_ptr, nbytes)):
            self.table[k] = self.table.get(k, 0) + c

    def export_size(self):
        return len(self._blob(self.table.items()))

    def export_to(self, dst_ptr, capacity):
        import ctypes
        b = self._blob(self.table.items())
        assert len(b) <= capacity
        ctypes.memmove(dst_ptr, b, len(b))
        return len(b)

    def wait_stream(self, stream=0):
        raise AssertionError("a CPU engine has no stream to wait for")


def _rank_table(rank, world):
    """keys shared by all ranks, by some, and private ones; sizes that differ per rank (one rank nearly empty)"""
    t = {}
    for i in range(40 + 300 * rank if rank != 1 else 3):
        t[b"shared-%d" % (i % 97)] = t.get(b"shared-%d" % (i % 97), 0) + i + 1
    for i in range(25 * (rank + 1)):
        t[b"rank%d-" % rank + bytes([65 + i % 26]) * (1 + i % 61)] = 1000 * rank + i
    t[b""] = rank + 1                                   # an empty key is a key
    return t


def _worker(rank, world, port, dst, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfind_amd.dist import exchange_shares
        eng = _DictEngine(_rank_table(rank, world))
        exchange_shares(eng, None, dst)
        want = {}
        for r in range(world):
            for k, c in _rank_table(r, world).items():
                want[k] = want.get(k, 0) + c
        if rank == dst:
            ok = eng.table == want
        else:                                           # every other rank keeps its own share, fully summed
            ok = eng.table == {k: c for k, c in want.items() if _DictEngine._owner(k, world) == rank}
        # weak-scaling bookkeeping: the per-rank counts all-reduce to the global total
        lo, hi = shard_range(10_000_019, rank, world)
        t = torch.tensor([hi - lo], dtype=torch.int64)
        dist.all_reduce(t)
        ok = ok and int(t.item()) == 10_000_019
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, 0), (3, 2)])
def test_exchange_shares_over_gloo_with_a_cpu_engine(world, dst):
    """vfind_amd.dist.exchange_shares - the owner-partitioned exchange with torch.distributed as the transport, what
    Engine.allreduce_table falls back to and what the one-GPU rehearsal runs - with world 2 and 3 over gloo: bucket sizes,
    padded offsets, the two all_to_all_single rounds (splits towards `dst` only in the second) and the merge order, on a
    dict-backed engine.  Root ends with the sum of all tables, every other rank with exactly its own share."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dst, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {r: True for r in range(world)}


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
