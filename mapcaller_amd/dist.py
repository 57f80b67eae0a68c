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


def finalize_planes(planes: torch.Tensor, G: int, max_dup: int = 5) -> torch.Tensor:
    """Field widths of MappingRecord_t (reference src/structure.h:152-163) on summed counters in the planes' own layout
    (api.planes_alloc): A,C,G,T,multi_hit saturate at 4095, readCount at the duplicate cap; F1,R2,F2,R1 are 16-bit words already.
    (torch version of k_prof_finalize, for planes that are not attached to a Mapper.)"""
    from . import api
    multi, half = api.planes_parts(planes, G)
    multi.clamp_(max=4095)
    for slot, top in ((0, 4095), (1, 4095), (2, 4095), (3, 4095), (4, max_dup)):
        v = half[slot].to(torch.int32) & 0xFFFF  # (int16 storage: counters from 2^15 on read back negative)
        half[slot] = v.clamp(max=top).to(torch.int16)
    return planes


def reduce_profile(planes: torch.Tensor, sparse, G: int, root: int = 0, shared_read_count: bool = True, mapper=None, via: str = "auto"):
    """Sums the counter planes of all ranks (api.planes_alloc's layout, csrc/mcx_planes.h: multi_hit as u32, the other nine as u16)
    onto ``root`` (RCCL reduce on GPU tensors; only the rank that calls the variants needs the sum), in pieces of 2^28 words
    (1 GiB: link speed, and no collective's count outgrows 32 bits), and gathers the sparse records of every rank in rank order.
    torch.distributed sums no 16-bit integers either, so the 16-bit planes travel as the int32 words they lie in, two positions
    to a word: A C G T (and readCount, when it is summed at all) clamped on every rank first — to 4095, up to 16 ranks cannot
    carry into the neighbouring half, and the clamped sum finalises to the same value —, F1 R2 F2 R1 as they are when no half can
    carry (the largest one over all ranks times the world size stays below 2^16; otherwise one counter per word, piece by piece):
    20 bytes per position on the wire.  ``shared_read_count``: the readCount plane already holds the run's count on every rank
    (the duplicate cap was decided across shards, mcx_batch_accumulate) and is left out of the sum; pass False for planes
    accumulated by independent runs.  What the root then holds equals the plain sum once finalised; the planes of the other
    ranks are scratch.  ``sparse`` is either the list of tuples of Mapper.profile_sparse() or the raw uint8 [n, 64] array of
    Mapper.profile_sparse_raw(); the same kind comes back.  Call after Mapper.profile_settle() and before finalisation — or pass
    the ``mapper`` that accumulated the planes and it is settled here first (idempotent): planes that still hold differences
    would be summed into plausible-looking garbage without any error.  With the gloo backend (CPU tests, several ranks on one
    GPU) device tensors are staged through the host.  ``reduce_profile.last_bytes``: what this rank put on the wire for the planes.
    ``via``: "reduce" — one reduce onto the root per piece (a ring or tree of RCCL's choosing: every byte of the planes crosses the
    root's ONE inbound link of that ring) — or "scatter": every piece cut into one slice per rank, the slices exchanged all to all
    (xGMI is a full mesh of point-to-point links: all of a GPU's links carry a slice at once), summed by their owners and gathered
    onto the root over all of its links — 2 x planes / world per link instead of planes; "auto": "scatter" from three ranks on."""
    from . import api
    reduce_profile.last_bytes = 0
    if mapper is not None:
        mapper.profile_settle()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return planes, (sparse if isinstance(sparse, np.ndarray) else list(sparse))
    world = dist.get_world_size()
    if world > 16:
        raise ValueError("reduce_profile: more than 16 ranks (A C G T travel as 16-bit halves clamped to 4095)")
    is_root = dist.get_rank() == root
    staged = dist.get_backend() != "nccl" and planes.is_cuda
    step = 1 << 28
    st = api.planes_stride(G)
    multi, half = api.planes_parts(planes, G)
    words = planes[st:].reshape(9, st // 2)  # the 16-bit planes as the int32 words they lie in

    scatter = via == "scatter" or (via == "auto" and world > 2)
    rank = dist.get_rank()

    def reduce_words(t):
        for lo in range(0, t.numel(), step):
            piece = t[lo:lo + step]
            if scatter:
                # slice j of the piece goes to rank j; every rank sums the slices it receives and the owners' sums are gathered onto the root
                n = piece.numel()
                per = (n + world - 1) // world
                send = torch.zeros(world * per, dtype=piece.dtype, device="cpu" if staged else piece.device)
                send[:n] = piece.cpu() if staged else piece
                recv = torch.empty_like(send)
                dist.all_to_all_single(recv, send)
                mine = recv.view(world, per).sum(0, dtype=torch.int32)
                parts = [torch.empty_like(mine) for _ in range(world)] if is_root else None
                dist.gather(mine, parts, dst=root)
                if is_root:
                    piece.copy_(torch.cat(parts)[:n])
                reduce_profile.last_bytes += (world - 1) * per * 4 + (0 if is_root else per * 4)
                continue
            if staged:
                h = piece.cpu()
                dist.reduce(h, dst=root, op=dist.ReduceOp.SUM)
                if is_root:
                    piece.copy_(h)
            else:
                dist.reduce(piece, dst=root, op=dist.ReduceOp.SUM)
            reduce_profile.last_bytes += piece.numel() * 4

    def clamp_halves(slot, top):  # in pieces: a genome-sized temporary may not fit beside the planes
        for lo in range(0, st, step):
            h = half[slot, lo:lo + step]
            v = h.to(torch.int32) & 0xFFFF
            h.copy_(v.clamp(max=top).to(torch.int16))

    for slot in (0, 1, 2, 3):
        clamp_halves(slot, 4095)
    reduce_words(words[0:4].reshape(-1))
    if not shared_read_count:
        clamp_halves(4, 4095)
        reduce_words(words[4])
    top = torch.zeros(1, dtype=torch.int64, device=planes.device)
    for slot in (5, 6, 7, 8):
        for lo in range(0, st, step):
            top = torch.maximum(top, (half[slot, lo:lo + step].to(torch.int32) & 0xFFFF).max().to(torch.int64).reshape(1))
    if staged:
        top = top.cpu()
    dist.all_reduce(top, op=dist.ReduceOp.MAX)
    if int(top.item()) * world <= 0xFFFF:
        reduce_words(words[5:9].reshape(-1))
    else:  # (sums of independent deep runs: one counter per word on the wire)
        for slot in (5, 6, 7, 8):
            for lo in range(0, st, step):
                h = half[slot, lo:lo + step]
                wide = (h.to(torch.int32) & 0xFFFF).contiguous()
                reduce_words(wide)
                if is_root:
                    v = wide & 0xFFFF
                    h.copy_(torch.where(v >= 0x8000, v - 0x10000, v).to(torch.int16))
    reduce_words(multi)
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
