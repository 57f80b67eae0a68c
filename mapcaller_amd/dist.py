"""Multi-GPU glue: one process per GPU, reads sharded, one exchange at the end of a run.

Reads shard embarrassingly: every rank maps a contiguous range of pairs against
its own replica of the index, so the data path has no collective.  The only exchange is the one
the reference's design implies when -vcf is on: the per-position counters that UpdateProfile
accumulates (reference src/AlignmentProfile.cpp:41-271) must be summed before VariantCalling
reads them, and the sparse tallies (insert / delete strings, break points, discordant sites)
concatenated.  With torch.distributed's "nccl" backend that all-reduce is RCCL over xGMI; the
same code runs on gloo for the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

CHUNK_PAIRS = 100  # ReadChunkSize / 2: shards start on the reference's chunk boundaries


def shard_pairs(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) range of pairs for `rank`; boundaries are multiples of 100 pairs so
    that every shard replays the reference's per-chunk insert-size feedback like a run of its own."""
    chunks = (n_pairs + CHUNK_PAIRS - 1) // CHUNK_PAIRS
    lo = (chunks * rank) // world * CHUNK_PAIRS
    hi = (chunks * (rank + 1)) // world * CHUNK_PAIRS
    return min(lo, n_pairs), min(hi, n_pairs)


def finalize_planes(planes: torch.Tensor, max_dup: int = 5) -> torch.Tensor:
    """Field widths of MappingRecord_t (reference src/structure.h:152-163) on summed counters:
    A,C,G,T,multi_hit saturate at 4095, readCount at the duplicate cap, F1,R2,F2,R1 wrap at 2^16.
    (torch version of k_prof_finalize, for tensors that are not attached to a Mapper.)"""
    planes[0:5].clamp_(max=4095)
    planes[5].clamp_(max=max_dup)
    planes[6:10].bitwise_and_(0xFFFF)
    return planes


def reduce_profile(planes: torch.Tensor, sparse):
    """Sums the [10, G] counter planes over all ranks in place (all-reduce: RCCL on GPU tensors),
    in pieces of 2^30 elements so that no collective's count outgrows 32 bits, and gathers the sparse
    records of every rank in rank order.  ``sparse`` is either the list of tuples of
    Mapper.profile_sparse() or the raw uint8 [n, 64] array of Mapper.profile_sparse_raw(); the same
    kind comes back.  Call before finalisation."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return planes, (sparse if isinstance(sparse, np.ndarray) else list(sparse))
    step = 1 << 30  # elements per collective: a GRCh38-sized plane has 3.1e9, more than a 32-bit count holds
    for k in range(planes.shape[0]):
        row = planes[k]
        for lo in range(0, row.numel(), step):
            dist.all_reduce(row[lo:lo + step], op=dist.ReduceOp.SUM)
    world = dist.get_world_size()
    if isinstance(sparse, np.ndarray):  # raw records: one padded all-gather of bytes
        dev = planes.device
        n = torch.tensor([sparse.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        most = max(int(c.item()) for c in counts)
        mine = torch.zeros((most, 64), dtype=torch.uint8, device=dev)
        if sparse.shape[0]:
            mine[: sparse.shape[0]] = torch.from_numpy(np.ascontiguousarray(sparse)).to(dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return planes, np.concatenate([p[: int(c.item())].cpu().numpy() for p, c in zip(parts, counts)], axis=0)
    parts = [None] * world
    dist.all_gather_object(parts, list(sparse))
    merged: List[tuple] = []
    for p in parts:
        merged.extend(p)
    return planes, merged


def sum_over_ranks(values: Sequence[int], device: torch.device) -> List[int]:
    """Run totals the variant caller needs from all shards (pairs, pair distance sum, read length sum)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
