"""The N > 1 path on CPU: world_size 2 over gloo.  Shards partition the pairs on chunk
boundaries, the profile reduce sums the planes onto the root (all but readCount, which every shard
already holds for the whole run) and concatenates the sparse tallies, the summed planes finalise to
the reference's field widths, and the per-round exchange of a file run gathers bytes in rank order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mapcaller_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(seed, G, top, strand_lo, strand_hi, read_count_top=3000):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randint(0, top, (10, G), generator=g, dtype=torch.int32)
    rows[6:10] = torch.randint(strand_lo, strand_hi, (4, G), generator=g, dtype=torch.int32)
    rows[5] = torch.randint(0, read_count_top, (G,), generator=g, dtype=torch.int32)
    return rows


def _expected(all_rows, root, shared_read_count, max_dup):
    """What the root's planes finalise to: the reference's field widths on the plain sum (readCount: the root's own unless it is summed)."""
    tot = sum(r.to(torch.int64) for r in all_rows)
    if shared_read_count:
        tot[5] = all_rows[root][5].to(torch.int64)
    tot[0:5].clamp_(max=4095)
    tot[5].clamp_(max=max_dup)
    tot[6:10] &= 0xFFFF
    return tot.to(torch.int32)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapcaller_amd import api
    G = 1000
    # strand depths near the 16-bit wrap: too large to travel as words of two positions
    planes = api.planes_from_rows(_rows(100 + rank, G, 3000, 40000, 43000, read_count_top=6))
    sparse = [("I", 10 * rank + k, "AC") for k in range(3)] + [("B", 7, "")]
    planes, merged = mdist.reduce_profile(planes, sparse, G, root=0)
    wire_a = mdist.reduce_profile.last_bytes
    # a second profile whose strand depths are small (they travel as words too) and whose counters pass 4095 on one rank, the readCount
    # plane summed as for independent runs: what the root gets equals the plain sum once the field widths are applied
    p2 = api.planes_from_rows(_rows(200 + rank, G, 9000, 0, 30000, read_count_top=6))
    p2, _ = mdist.reduce_profile(p2, [], G, root=0, shared_read_count=False)
    wire_b = mdist.reduce_profile.last_bytes
    # the dist_exchange the shards of a file run share: bytes of every rank in rank order, any length
    link = api.dist_exchange()
    for nbytes in (4, 1000, 0, 33):
        send = np.full(max(nbytes, 1), 7 * rank + 1, dtype=np.uint8)
        recv = np.zeros(max(world * nbytes, 1), dtype=np.uint8)
        assert link.allgather(None, send.ctypes.data, recv.ctypes.data, nbytes) == 0
        if nbytes:
            assert recv.reshape(world, nbytes)[:, 0].tolist() == [7 * q + 1 for q in range(world)]
    raw = np.full((2 + rank, 64), rank + 1, dtype=np.uint8)  # raw record arrays of different lengths
    _, raw_all = mdist.reduce_profile(api.planes_alloc(4, "cpu"), raw, 4)
    assert raw_all.shape == (5, 64) and raw_all[:2].max() == 1 and raw_all[2:].min() == 2
    lo, hi = mdist.shard_pairs(12345, rank, world)
    t = mdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    tot = mdist.sum_over_ranks([100 + rank, 7 * (rank + 1), 1 << 40], torch.device("cpu"))  # run totals for the variant caller
    if rank == 0:
        torch.save({"sum": planes, "merged": merged, "t": t, "tot": tot, "second": p2, "wire": (wire_a, wire_b)}, out)
    # every rank checks its own shard bounds
    assert lo % 100 == 0 and (hi % 100 == 0 or hi == 12345)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi))
    assert gathered[0][0] == 0 and gathered[-1][1] == 12345
    for a, b in zip(gathered, gathered[1:]):
        assert a[1] == b[0]
    dist.destroy_process_group()


def test_profile_reduce_and_sharding_world2(tmp_path):
    from mapcaller_amd import api
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    G, stride = 1000, api.planes_stride(1000)
    rows = [_rows(100 + q, G, 3000, 40000, 43000, read_count_top=6) for q in range(2)]
    got = api.planes_view(mdist.finalize_planes(r["sum"].clone(), G, max_dup=5), G)
    assert torch.equal(got, _expected(rows, 0, True, 5))  # (the readCount plane is the run's on every rank already: not summed)
    assert int((rows[0][6:10] + rows[1][6:10]).min()) > 0xFFFF  # (the strand counters did wrap)
    rows2 = [_rows(200 + q, G, 9000, 0, 30000, read_count_top=6) for q in range(2)]
    got2 = api.planes_view(mdist.finalize_planes(r["second"].clone(), G, max_dup=15), G)
    assert torch.equal(got2, _expected(rows2, 0, False, 15))
    # bytes on the wire per rank and position of the padded genome: A C G T 8, the strand planes 8 as words or 16 a counter per word,
    # multi_hit 4, readCount 2 when it is summed
    assert r["wire"] == ((8 + 16 + 4) * stride, (8 + 2 + 8 + 4) * stride)
    assert len(r["merged"]) == 8 and r["merged"][0] == ("I", 0, "AC") and r["merged"][4] == ("I", 10, "AC")
    assert r["t"] == 2.0
    assert r["tot"] == [201, 21, 1 << 41]


def test_shards_cover_everything_for_any_world():
    for n in (0, 1, 99, 100, 101, 12345, 2_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [mdist.shard_pairs(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo % 100 == 0 for lo, _ in spans)


def _rows3(rank, world, G):
    rows = _rows(300 + rank, G, 5000, 0, 65536 // world, read_count_top=6)  # halves that just cannot carry
    rows[6, 0] = 65535 // world                                              # (the largest value the guard lets through)
    return rows


def _worker3(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapcaller_amd import api
    G = 777
    got, _ = mdist.reduce_profile(api.planes_from_rows(_rows3(rank, world, G)), [], G, root=1, via="reduce")  # a root that is not rank 0
    wire = mdist.reduce_profile.last_bytes
    # the same sum with the pieces scattered over the ranks, summed by their owners and gathered onto the root (what three ranks and more take)
    got2, _ = mdist.reduce_profile(api.planes_from_rows(_rows3(rank, world, G)), [], G, root=1)
    wire2 = mdist.reduce_profile.last_bytes
    if rank == 1:
        torch.save({"got": got, "wire": wire, "scattered": got2, "wire_scattered": wire2}, out)
    dist.destroy_process_group()


def test_reduce_world3_root1(tmp_path):
    """Three ranks, the sum lands on rank 1, strand depths at the edge of what may travel two to a word: 20 bytes per position on the
    wire, and what the root holds equals the plain sum once the field widths are applied (readCount left alone)."""
    from mapcaller_amd import api
    out = str(tmp_path / "r1.pt")
    mp.spawn(_worker3, args=(3, _free_port(), out), nprocs=3, join=True)
    r = torch.load(out)
    G = 777
    assert r["wire"] == 20 * api.planes_stride(G)
    got = api.planes_view(mdist.finalize_planes(r["got"].clone(), G, max_dup=15), G)
    assert torch.equal(got, _expected([_rows3(q, 3, G) for q in range(3)], 1, True, 15))
    # scattered: the root holds the same planes; it sent two thirds of every piece (and received as much, then the owners' sums)
    assert torch.equal(r["scattered"], r["got"])
    assert 0 < r["wire_scattered"] <= 20 * api.planes_stride(G) * 2 // 3 + 64
