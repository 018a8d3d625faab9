"""Multi-GPU plumbing for the one real exchange step of the path.

The graph is replicated on every GPU; the ascending source list is block-partitioned over ranks
(contiguous blocks, so concatenating the ranks' candidate lists in rank order *is* the replay order);
after the per-rank SSSP stage there is ONE variable-length all-gather of candidate lists
(``torch.distributed`` = RCCL over xGMI on GPUs, gloo in the CPU tests), then the sequential claim
replay. No other collective exists on the path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def partition_sources(n_sources: int, world_size: int) -> list[tuple[int, int]]:
    """Contiguous, near-equal blocks [lo, hi) per rank (the reference hands out contiguous chunks of
    out_nodes too, greedytigs/mod.rs:573-616, but dynamically; here the split is static)."""
    base, rem = divmod(n_sources, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def partition_sources_by_work(work: np.ndarray, world_size: int) -> list[tuple[int, int]]:
    """Contiguous blocks [lo, hi) of near-equal cumulative estimated work (e.g. 1 + out-degree of each source,
    in source order) so that ranks finish their SSSP stage together. Deterministic: every rank computes the same split."""
    n = int(len(work))
    if n == 0:
        return [(0, 0)] * world_size
    cum = np.cumsum(np.asarray(work, dtype=np.int64))
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = (total * r) // world_size
        cuts.append(max(cuts[-1], int(np.searchsorted(cum, target, side="left"))))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def allgather_candidates(start: torch.Tensor, count: torch.Tensor, pool: torch.Tensor, pool_used: int,
                         ranges: list[tuple[int, int]], group=None):
    """All ranks contribute (start[int64 n_r], count[int32 n_r], pool[int64 >= pool_used]) for their source block.

    Returns (start_all, count_all, pool_all) on the same device, indexed by absolute source index, with ``start`` rebased
    into the concatenated pool. The all-gather is variable-length, so it is issued as one broadcast per (rank, array) at the
    EXACT sizes, straight into the slices of the concatenated result (grouped broadcasts; on GPUs RCCL runs them
    back to back over xGMI): no padding to the largest rank and no per-rank staging copies -- the receive buffers ARE the
    result. One tiny all-gather of the pool sizes precedes it.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = pool.device
    sizes = torch.tensor([pool_used], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    pool_sizes = [int(t.item()) for t in all_sizes]
    n_total = ranges[-1][1]
    start_all = torch.empty(n_total, dtype=torch.int64, device=dev)
    count_all = torch.empty(n_total, dtype=torch.int32, device=dev)
    pool_all = torch.empty(sum(pool_sizes), dtype=torch.int64, device=dev)
    lo, hi = ranges[rank]
    off = sum(pool_sizes[:rank])
    # own block: rebased before it travels. The engine leaves the start of an EMPTY list untouched (mtg_sssp_candidates): such
    # words are set to 0 here, so that nothing uninitialised is broadcast and every start indexes the concatenated pool.
    start_all[lo:hi] = torch.where(count[: hi - lo] > 0, start[: hi - lo] + off, torch.zeros_like(start[: hi - lo]))
    count_all[lo:hi] = count[: hi - lo]
    pool_all[off:off + pool_used] = pool[:pool_used]
    pending = []
    off = 0
    for r, (lo, hi) in enumerate(ranges):
        src = dist.get_global_rank(group, r) if group is not None else r
        for view in (start_all[lo:hi], count_all[lo:hi], pool_all[off:off + pool_sizes[r]]):
            if view.numel():
                pending.append(dist.broadcast(view, src=src, group=group, async_op=True))
        off += pool_sizes[r]
    for w in pending:
        w.wait()
    return start_all, count_all, pool_all


def to_numpy_u(start_all: torch.Tensor, count_all: torch.Tensor, pool_all: torch.Tensor):
    return (start_all.cpu().numpy().view(np.uint64), count_all.cpu().numpy().view(np.uint32),
            pool_all.cpu().numpy().view(np.uint64))
