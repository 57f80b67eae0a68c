"""Multi-GPU glue: one process per GPU, reads sharded, one bulk exchange at the end of a run.

Reads shard embarrassingly: every rank maps its batches against its own replica of
the index.  What the ranks exchange while they map is a few KB per round (per-chunk pair sums for
the one insert-size trajectory of the stream, duplicate-check keys: api.dist_exchange, handled
inside mcx_map_files_ex).  The bulk exchange is the one the reference's design implies when -vcf is
on: the per-position counters that UpdateProfile accumulates (reference
src/AlignmentProfile.cpp:41-271) must be summed before VariantCalling reads them, and the sparse
tallies (insert / delete strings, break points, discordant-pair events) concatenated.  With
torch.distributed's "nccl" backend that reduce is RCCL over xGMI; the same code runs on gloo for the
CPU tests.
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


READ_COUNT_PLANE = 5


def reduce_profile(planes: torch.Tensor, sparse, root: int = 0, shared_read_count: bool = True):
    """Sums the [10, G] counter planes of all ranks onto ``root`` (RCCL reduce on GPU tensors; only the
    rank that calls the variants needs the sum), in pieces of 2^28 elements (1 GiB: link speed, and no
    collective's count outgrows 32 bits), and gathers the sparse records of every rank in rank order.
    ``shared_read_count``: the readCount plane already holds the run's count on every rank (the
    duplicate cap was decided across shards, mcx_batch_accumulate) and is left out of the sum; pass
    False for planes accumulated by independent runs.  ``sparse`` is either the list of tuples of
    Mapper.profile_sparse() or the raw uint8 [n, 64] array of Mapper.profile_sparse_raw(); the same
    kind comes back.  Call before finalisation.  With the gloo backend (CPU tests, several ranks on
    one GPU) device tensors are staged through the host."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return planes, (sparse if isinstance(sparse, np.ndarray) else list(sparse))
    staged = dist.get_backend() != "nccl" and planes.is_cuda
    step = 1 << 28
    for k in range(planes.shape[0]):
        if shared_read_count and k == READ_COUNT_PLANE:
            continue
        row = planes[k]
        for lo in range(0, row.numel(), step):
            piece = row[lo:lo + step]
            if staged:
                h = piece.cpu()
                dist.reduce(h, dst=root, op=dist.ReduceOp.SUM)
                if dist.get_rank() == root:
                    piece.copy_(h)
            else:
                dist.reduce(piece, dst=root, op=dist.ReduceOp.SUM)
    world = dist.get_world_size()
    if isinstance(sparse, np.ndarray):  # raw records: one padded all-gather of bytes
        dev = planes.device if not staged else torch.device("cpu")
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


def _coll_device(device: torch.device) -> torch.device:
    """Small collectives run on the device with RCCL and on the host with gloo."""
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def sum_over_ranks(values: Sequence[int], device: torch.device) -> List[int]:
    """Run totals the variant caller needs from all shards (pairs, pair distance sum, read length sum)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
