"""Experiment: a context created after another one was closed maps ~1 ms per step slower.  A: first context; B: a second one
while A is still open; C: a third after A and B were closed.  Measured: A and B 20.9-21.8 ms in either order, C 22.2-22.5
(k_pack_reads 0.7 -> 1.0 ms) — whatever the cause, it is the closing and re-creating, not being second; 96 GB of ballast taken
before A (so that A's memory is not what the index build gave back) changes nothing.   python scripts/ctx_order.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mapcaller_amd import api


def main():
    args = bench.parse()
    dev = torch.device("cuda", 0)
    codes, lens, _ = bench.make_genome(args, dev, seed=1234)
    index = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=int(args.full_sa))
    n = 2 * args.batch_pairs
    batches = [bench.make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1000 + s, device=dev).reshape(-1).contiguous() for s in range(4)]
    del codes
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    aln = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    cig = torch.empty(api.cigar_pool_words(n), dtype=torch.int32, device=dev)

    def run(mp):
        mp.map_batch_dev(batches[0].data_ptr(), off.data_ptr(), n, True, aln.data_ptr(), cig.data_ptr())
        before = mp.stats.as_dict()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for b in batches[1:]:
            mp.map_batch_dev(b.data_ptr(), off.data_ptr(), n, True, aln.data_ptr(), cig.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        after = mp.stats.as_dict()
        return {"ms_per_step": round(1000 * dt, 3), "encode": round((after["ms_encode"] - before["ms_encode"]) / 3, 3), "rescue": round((after["ms_rescue"] - before["ms_rescue"]) / 3, 3)}

    mk = lambda: api.Mapper(index, alg=args.alg, max_read_len=256, max_batch_reads=n)
    out = {}
    a = mk(); out["A first"] = run(a)
    b = mk(); out["B second, A open"] = run(b)
    out["A again, B open"] = run(a)
    a.close(); b.close()
    c = mk(); out["C after both closed"] = run(c)
    c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
