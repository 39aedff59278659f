"""Multi-GPU plumbing over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The hot path shards by reads with no data-path collective; the only exchange is the final
(key,count) table merge (the cross-GPU half of the reduce at src/lib.rs:292-306)."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) share of n_items for `rank`; shares differ by at most one item"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _padded(n: int) -> int:
    return (n + 255) // 256 * 256


def exchange_shares(engine, group=None, dst: int = 0):
    """The owner-partitioned reduce of the count tables with torch.distributed as the transport: every rank
    buckets its entries by owner (vf_table_partition_export), ONE all_to_all_single takes bucket o of every rank to rank o
    and nowhere else (every entry crosses the wire once, as in the library's own RCCL exchange), every owner clears its
    table and merges the buckets it holds, and a second all_to_all_single - all splits zero except towards `dst` - takes
    the owners' tables (disjoint key sets) to `dst`, which merges them.  Works on nccl (device buffers) and on gloo (the
    CPU-side rehearsal: the bytes are staged through host memory).

    `engine` is anything with the Engine methods used here (partition_sizes / partition_export / clear / merge_from /
    export_size / export_to / wait_stream on raw addresses).  An engine with a `torch_device` attribute keeps its blobs on
    that device - tests/test_distributed_cpu.py runs this function's split / offset arithmetic over gloo with a dict-backed
    stand-in on CPU tensors, no GPU involved."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = getattr(engine, "torch_device", None) or torch.device("cuda", engine.device)
    on_gpu = dev.type == "cuda"
    via_host = dist.get_backend(group) == "gloo" and on_gpu
    wire = torch.device("cpu") if via_host else dev

    def lib_waits_for_torch():
        # buffers torch has just written go to the library only behind an event on torch's stream (vf_wait_stream); the
        # other direction needs nothing: vf_table_export / _merge return when the library's stream is done with the buffer
        if on_gpu:
            engine.wait_stream(torch.cuda.current_stream(dev).cuda_stream)

    def all_sizes(mine):                                         # [from][to], bytes
        t = torch.tensor(mine, dtype=torch.int64, device=wire)
        out = torch.empty(world * world, dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(out, t, group=group)
        return out.cpu().view(world, world).tolist()

    def move(send, in_splits, out_splits):                       # device buffer -> device buffer through the group
        recv = torch.empty(max(sum(out_splits), 1), dtype=torch.uint8, device=dev)[:sum(out_splits)]
        s_w = send[:sum(in_splits)].to(wire) if via_host else send[:sum(in_splits)]
        r_w = torch.empty(sum(out_splits), dtype=torch.uint8, device=wire) if via_host else recv
        dist.all_to_all_single(r_w, s_w, out_splits, in_splits, group=group)
        if via_host:
            recv.copy_(r_w)
        return recv

    # buffers are allocated without a fill kernel (a torch.zeros still queued on torch's stream once landed on top of the
    # exported bytes: lost entries on a cold GPU)
    sizes = engine.partition_sizes(world)
    pad = [_padded(x) for x in sizes]
    send = torch.empty(max(sum(pad), 256), dtype=torch.uint8, device=dev)
    lib_waits_for_torch()
    at = 0
    for owner in range(world):
        engine.partition_export(world, owner, send.data_ptr() + at, pad[owner])
        at += pad[owner]
    table = all_sizes(sizes)
    out_pad = [_padded(table[q][rank]) for q in range(world)]
    recv = move(send, pad, out_pad)
    lib_waits_for_torch()
    engine.clear()
    at = 0
    for q in range(world):
        engine.merge_from(recv.data_ptr() + at, table[q][rank])
        at += out_pad[q]
    # shares to dst
    mine = engine.export_size()
    shares = all_sizes([mine if q == dst and rank != dst else 0 for q in range(world)])
    send = torch.empty(max(_padded(mine), 256), dtype=torch.uint8, device=dev)
    lib_waits_for_torch()
    if rank != dst:
        engine.export_to(send.data_ptr(), send.numel())
    in_splits = [(_padded(mine) if q == dst and rank != dst else 0) for q in range(world)]
    out_splits = [(_padded(shares[q][dst]) if rank == dst else 0) for q in range(world)]
    recv = move(send, in_splits, out_splits)
    if rank == dst:
        lib_waits_for_torch()
        at = 0
        for q in range(world):
            if q != rank:
                engine.merge_from(recv.data_ptr() + at, shares[q][dst])
            at += out_splits[q]


def allreduce_local(engines, root: int = 0):
    """The same exchange between contexts of ONE process (several GPUs driven by one host thread, or a one-GPU rehearsal
    of the partition logic): bucket o of every context goes to context o by device memory, no collective involved."""
    import torch

    world = len(engines)
    if world == 1:
        return
    buckets = []
    for e in engines:
        sizes = e.partition_sizes(world)
        row = []
        for owner in range(world):
            buf = torch.empty(max(sizes[owner], 16), dtype=torch.uint8, device=torch.device("cuda", e.device))
            e.partition_export(world, owner, buf.data_ptr(), buf.numel())
            row.append((buf, sizes[owner]))
        buckets.append(row)
    for o, e in enumerate(engines):
        e.clear()
        dev = torch.device("cuda", e.device)
        for q in range(world):
            buf, nbytes = buckets[q][o]
            here = buf if buf.device == dev else buf.to(dev)        # kept alive across the (synchronous) merge
            if here is not buf:
                torch.cuda.synchronize(dev)                         # the copy ran on torch's stream, the merge will not
            e.merge_from(here.data_ptr(), nbytes)
    for o, e in enumerate(engines):
        if o == root:
            continue
        n = e.export_size()
        buf = torch.empty(n + 64, dtype=torch.uint8, device=torch.device("cuda", e.device))
        e.export_to(buf.data_ptr(), buf.numel())
        rdev = torch.device("cuda", engines[root].device)
        b2 = buf.to(rdev) if buf.device != rdev else buf
        if b2 is not buf:
            torch.cuda.synchronize(rdev)
        engines[root].merge_from(b2.data_ptr(), n)
