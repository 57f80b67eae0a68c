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


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 1000
    g = torch.Generator().manual_seed(100 + rank)
    planes = torch.randint(0, 3000, (10, G), generator=g, dtype=torch.int32)
    planes[6:10] += 40000  # strand depths near the 16-bit wrap
    mine = planes.clone()
    sparse = [("I", 10 * rank + k, "AC") for k in range(3)] + [("B", 7, "")]
    planes, merged = mdist.reduce_profile(planes, sparse, root=0)
    wire_a = mdist.reduce_profile.last_bytes  # strand depths too large to share a word: A|C, G|T packed, the other five alone
    # a second profile whose strand depths are small (they share words too) and whose counters pass 4095 on one rank:
    # what the root gets equals the plain sum once the field widths are applied
    g2 = torch.Generator().manual_seed(200 + rank)
    p2 = torch.randint(0, 9000, (10, G), generator=g2, dtype=torch.int32)
    p2[6:10] = torch.randint(0, 30000, (4, G), generator=g2, dtype=torch.int32)
    p2[5] = torch.randint(0, 6, (G,), generator=g2, dtype=torch.int32)
    mine2 = p2.clone()
    p2, _ = mdist.reduce_profile(p2, [], root=0, shared_read_count=False)
    wire_b = mdist.reduce_profile.last_bytes
    p3 = mine2.clone()
    p3, _ = mdist.reduce_profile(p3, [], root=0, shared_read_count=False, packed=False)
    wire_c = mdist.reduce_profile.last_bytes
    # the dist_exchange the shards of a file run share: bytes of every rank in rank order, any length
    from mapcaller_amd import api
    link = api.dist_exchange()
    for nbytes in (4, 1000, 0, 33):
        send = np.full(max(nbytes, 1), 7 * rank + 1, dtype=np.uint8)
        recv = np.zeros(max(world * nbytes, 1), dtype=np.uint8)
        assert link.allgather(None, send.ctypes.data, recv.ctypes.data, nbytes) == 0
        if nbytes:
            assert recv.reshape(world, nbytes)[:, 0].tolist() == [7 * q + 1 for q in range(world)]
    raw = np.full((2 + rank, 64), rank + 1, dtype=np.uint8)  # raw record arrays of different lengths
    _, raw_all = mdist.reduce_profile(torch.zeros((10, 4), dtype=torch.int32), raw)
    assert raw_all.shape == (5, 64) and raw_all[:2].max() == 1 and raw_all[2:].min() == 2
    lo, hi = mdist.shard_pairs(12345, rank, world)
    t = mdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    tot = mdist.sum_over_ranks([100 + rank, 7 * (rank + 1), 1 << 40], torch.device("cpu"))  # run totals for the variant caller
    if rank == 0:
        torch.save({"sum": planes, "mine": mine, "merged": merged, "t": t, "tot": tot, "packed": p2, "plain": p3, "wire": (wire_a, wire_b, wire_c)}, out)
    # every rank checks its own shard bounds
    assert lo % 100 == 0 and (hi % 100 == 0 or hi == 12345)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi))
    assert gathered[0][0] == 0 and gathered[-1][1] == 12345
    for a, b in zip(gathered, gathered[1:]):
        assert a[1] == b[0]
    dist.destroy_process_group()


def test_profile_reduce_and_sharding_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    g1 = torch.Generator().manual_seed(101)
    other = torch.randint(0, 3000, (10, 1000), generator=g1, dtype=torch.int32)
    other[6:10] += 40000
    want = r["mine"] + other
    want[5] = r["mine"][5]  # the readCount plane is the run's on every rank already: not summed
    assert torch.equal(r["sum"], want)
    fin = mdist.finalize_planes(want.clone(), max_dup=5)
    assert int(fin[0:5].max()) <= 4095 and int(fin[5].max()) <= 5 and int(fin[6:10].max()) <= 0xFFFF
    assert torch.equal(fin[6], (want[6] & 0xFFFF))
    assert r["wire"] == (7 * 4000, 5 * 4000, 10 * 4000)  # bytes on the wire per rank: 7, 5 and 10 planes of 1000 u32
    assert not torch.equal(r["packed"], r["plain"])  # (counters above 4095 were clamped before they travelled)
    assert torch.equal(mdist.finalize_planes(r["packed"].clone(), max_dup=15), mdist.finalize_planes(r["plain"].clone(), max_dup=15))
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


def _worker3(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 777
    g = torch.Generator().manual_seed(300 + rank)
    p = torch.randint(0, 5000, (10, G), generator=g, dtype=torch.int32)
    p[6:10] = torch.randint(0, 65536 // world, (4, G), generator=g, dtype=torch.int32)  # low halves that just cannot carry
    p[6, 0] = 65535 // world                                                             # (the largest value the guard lets through)
    mine = p.clone()
    got, _ = mdist.reduce_profile(p, [], root=1)  # a root that is not rank 0
    wire = mdist.reduce_profile.last_bytes
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 1:
        torch.save({"got": got, "all": gathered, "wire": wire}, out)
    dist.destroy_process_group()


def test_packed_reduce_world3_root1(tmp_path):
    """Three ranks, the sum lands on rank 1, strand depths at the edge of what may share a word: five planes on the wire, and what
    the root holds equals the plain sum once the field widths are applied (readCount left alone)."""
    out = str(tmp_path / "r1.pt")
    mp.spawn(_worker3, args=(3, _free_port(), out), nprocs=3, join=True)
    r = torch.load(out)
    want = r["all"][0] + r["all"][1] + r["all"][2]
    want[5] = r["all"][1][5]
    assert r["wire"] == 5 * 777 * 4
    assert torch.equal(mdist.finalize_planes(r["got"].clone(), max_dup=15), mdist.finalize_planes(want.clone(), max_dup=15))
