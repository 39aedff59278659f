"""Multi-GPU plumbing over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The hot path shards by reads with no data-path collective; the only exchange is the final
(key,count) table merge (the cross-GPU half of the reduce at src/lib.rs:292-306)."""
from __future__ import annotations

from typing import List, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) share of n_items for `rank`; shares differ by at most one item"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_blobs(send, nbytes: int, group=None) -> List[Tuple[object, int]]:
    """All-gather variable-size byte blobs.

    `send` is a uint8 tensor (any device) whose first `nbytes` bytes are this rank's blob.  Returns,
    per rank, (uint8 tensor view, valid byte count).  Blobs are padded to the largest size (rounded
    up to 256 B) so a single all_gather_into_tensor moves everything (one collective, large
    messages: xGMI rings are per-link bound, so fewer/larger is better than per-peer sends)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # gloo (CPU tests, or a single-GPU rehearsal of the multi-rank path) moves the bytes through
    # host memory; nccl (= RCCL) gathers device to device over xGMI
    via_host = dist.get_backend(group) == "gloo" and send.device.type != "cpu"
    wire = torch.device("cpu") if via_host else send.device
    sizes = torch.zeros(world, dtype=torch.int64, device=wire)
    sizes[rank] = nbytes
    dist.all_reduce(sizes, group=group)
    sizes_h = [int(x) for x in sizes.cpu().tolist()]
    cap = (max(sizes_h) + 255) // 256 * 256
    buf = torch.zeros(cap, dtype=torch.uint8, device=wire)
    buf[:nbytes] = send[:nbytes].to(wire)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=wire)
    dist.all_gather_into_tensor(recv, buf, group=group)
    if via_host:
        recv = recv.to(send.device)
    return [(recv[r * cap:(r + 1) * cap], sizes_h[r]) for r in range(world)]
