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


# two 16-bit counters share a u32 on the wire: (low plane, high plane)
PACKED_SATURATING = ((0, 1), (2, 3))   # A|C, G|T: every rank clamps to 4095 first, so 16 ranks cannot carry into the high half
PACKED_WITH_READ_COUNT = (4, 5)        # multi_hit | readCount, when the readCount plane is summed at all
PACKED_WRAPPING = ((6, 7), (8, 9))     # F1|R2, F2|R1: 16-bit fields that wrap; packed when the low halves cannot carry


def reduce_profile(planes: torch.Tensor, sparse, root: int = 0, shared_read_count: bool = True, packed: bool = True, mapper=None):
    """Sums the [10, G] counter planes of all ranks onto ``root`` (RCCL reduce on GPU tensors; only the
    rank that calls the variants needs the sum), in pieces of 2^28 elements (1 GiB: link speed, and no
    collective's count outgrows 32 bits), and gathers the sparse records of every rank in rank order.
    ``shared_read_count``: the readCount plane already holds the run's count on every rank (the
    duplicate cap was decided across shards, mcx_batch_accumulate) and is left out of the sum; pass
    False for planes accumulated by independent runs.  ``packed``: the counters are 12- and 16-bit fields
    (finalize_planes), so two planes travel in one u32 — A|C, G|T and multi_hit|readCount clamped to 4095
    on every rank first (up to 16 ranks: no carry), F1|R2 and F2|R1 when no low half can carry (the largest
    low-half value over all ranks times the world size stays below 2^16; otherwise those four travel alone)
    — five planes on the wire instead of nine or ten.  What the root then holds equals the plain sum once
    finalised; the planes of the other ranks are left in their packed form.  ``sparse`` is either the list
    of tuples of Mapper.profile_sparse() or the raw uint8 [n, 64] array of Mapper.profile_sparse_raw(); the
    same kind comes back.  Call after Mapper.profile_settle() and before finalisation — or pass the ``mapper`` that
    accumulated the planes and it is settled here first (idempotent): planes that still hold differences would be
    packed into plausible-looking garbage without any error.  With the gloo
    backend (CPU tests, several ranks on one GPU) device tensors are staged through the host.
    ``reduce_profile.last_bytes``: what this rank put on the wire for the planes."""
    reduce_profile.last_bytes = 0
    if mapper is not None:
        mapper.profile_settle()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return planes, (sparse if isinstance(sparse, np.ndarray) else list(sparse))
    world = dist.get_world_size()
    is_root = dist.get_rank() == root
    staged = dist.get_backend() != "nccl" and planes.is_cuda
    step = 1 << 28
    G = planes.shape[1]

    def reduce_row(k):
        row = planes[k]
        for lo in range(0, G, step):
            piece = row[lo:lo + step]
            if staged:
                h = piece.cpu()
                dist.reduce(h, dst=root, op=dist.ReduceOp.SUM)
                if is_root:
                    piece.copy_(h)
            else:
                dist.reduce(piece, dst=root, op=dist.ReduceOp.SUM)
            reduce_profile.last_bytes += piece.numel() * 4

    def pack(a, b, clamp):  # in pieces: a genome-sized temporary may not fit beside the planes
        for lo in range(0, G, step):
            x, y = planes[a, lo:lo + step], planes[b, lo:lo + step]
            if clamp:
                x.clamp_(max=4095)
                x.add_(y.clamp(max=4095) << 16)
            else:
                x.bitwise_and_(0xFFFF)
                x.bitwise_or_(y << 16)

    def unpack(a, b):
        for lo in range(0, G, step):
            x, y = planes[a, lo:lo + step], planes[b, lo:lo + step]
            y.copy_((x >> 16) & 0xFFFF)
            x.bitwise_and_(0xFFFF)

    pairs, alone = [], []
    if packed and world <= 16:
        pairs += [(a, b, True) for a, b in PACKED_SATURATING]
        if shared_read_count:
            alone.append(4)
        else:
            pairs.append(PACKED_WITH_READ_COUNT + (True,))
    else:
        alone += [0, 1, 2, 3, 4] + ([] if shared_read_count else [READ_COUNT_PLANE])
    wrap_ok = False
    if packed:
        top = torch.zeros(1, dtype=torch.int64, device=planes.device)
        for a, _ in PACKED_WRAPPING:
            for lo in range(0, G, step):
                top = torch.maximum(top, (planes[a, lo:lo + step] & 0xFFFF).max().to(torch.int64).reshape(1))
        if staged:
            top = top.cpu()
        dist.all_reduce(top, op=dist.ReduceOp.MAX)
        wrap_ok = int(top.item()) * world <= 0xFFFF
    if wrap_ok:
        pairs += [(a, b, False) for a, b in PACKED_WRAPPING]
    else:
        alone += [6, 7, 8, 9]
    for a, b, clamp in pairs:
        pack(a, b, clamp)
        reduce_row(a)
        if is_root:
            unpack(a, b)
    for k in alone:
        reduce_row(k)
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
