"""The -vcf bookkeeping of one bench batch, several times over: are the planes and the sparse tallies the same run against run?
    python scripts/determinism_vcf.py [--runs 3] [bench.py's workload flags]
Keeps the low byte of every finalized counter of run 0 (31 GB at 3.1 Gbp) and compares the later runs with it."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from mapcaller_amd import api


def main():
    runs = 3
    if "--runs" in sys.argv:
        k = sys.argv.index("--runs"); runs = int(sys.argv[k + 1]); del sys.argv[k:k + 2]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    codes, lens, _ = bench.make_genome(args, dev, seed=1234)
    index = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=int(args.full_sa))
    G = index.genome_size
    n = 2 * args.batch_pairs
    batch = bench.make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1001, device=dev, sub=args.sub, ins=args.ins, dele=args.dele, paired=True).reshape(-1).contiguous()
    del codes
    sl = min(n, args.vcf_slice_reads)
    off = (torch.arange(sl + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    import os
    if sl * 3072 > args.vcf_tier1_gb << 30:  # (as bench.py's -vcf leg: the large tier sized for this workload, so that everything fits beside the planes)
        os.environ["MCX_TIER1_GB"] = str(args.vcf_tier1_gb)
    mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=sl)
    os.environ.pop("MCX_TIER1_GB", None)
    planes = api.planes_alloc(G, dev)
    d_aln = torch.zeros(sl * 64, dtype=torch.uint8, device=dev)
    d_cig = torch.zeros(api.cigar_pool_words(sl), dtype=torch.int32, device=dev)
    starts = np.concatenate([[0], np.cumsum(lens)])
    keep, sparse0, alns = None, None, []
    res = {"reads": n, "runs": runs, "compare": []}
    for r in range(runs):
        planes.zero_()
        mapper.reset()
        mapper.profile_attach(planes.data_ptr())
        for lo in range(0, n, sl):
            m = min(sl, n - lo)
            mapper.map_batch_dev(batch.data_ptr() + lo * args.rlen, off.data_ptr(), m, True, d_aln.data_ptr(), d_cig.data_ptr())
            if r == 0:
                torch.cuda.synchronize()
                alns.append(d_aln.cpu().numpy().view(api.ALN_DTYPE)[:m].copy())
        mapper.profile_finalize(planes.data_ptr())
        sp = mapper.profile_sparse_raw().copy()
        sp[:, 10:] *= (np.arange(54)[None, :] < sp[:, 9:10]).astype(np.uint8)  # (the bytes behind a string are not part of the record)
        rows = set(bytes(x) for x in sp)
        def digest(lo, hi):  # one byte per position over the ten planes
            h = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
            v = api.planes_view(planes, G, lo, hi)
            for k in range(10):
                h += v[k] * (2 * k + 3)
            return (h ^ (h >> 8) ^ (h >> 16)).to(torch.uint8)
        if r == 0:
            keep = torch.empty(G, dtype=torch.uint8, device=dev)
            for lo in range(0, G, 1 << 28):
                hi = min(G, lo + (1 << 28))
                keep[lo:hi] = digest(lo, hi)
            sparse0 = rows
            continue
        out = {"run": r, "sparse_only_run0": len(sparse0 - rows), "sparse_only_this": len(rows - sparse0), "sparse_examples": []}
        for b in list(sparse0 - rows)[:5] + list(rows - sparse0)[:5]:
            out["sparse_examples"].append({"in": "run0" if b in sparse0 else "this", "pos": int(np.frombuffer(b[:8], dtype="<i8")[0]), "type": chr(b[8]), "len": b[9], "seq": b[10:10 + min(b[9], 20)].decode("latin1")})
        first, cnt = [], 0
        for lo in range(0, G, 1 << 28):
            hi = min(G, lo + (1 << 28))
            dif = (digest(lo, hi) != keep[lo:hi]).nonzero().flatten()
            cnt += dif.numel()
            for p in dif[:200:8].tolist():
                if len(first) < 24:
                    first.append({"pos": lo + p, "this": [int(x) for x in api.planes_view(planes, G, lo + p, lo + p + 1)[:, 0]]})
        out["differing_positions"] = cnt
        out["first"] = first
        # the reads of run 0 that lie over the first differing positions
        near = []
        aln = np.concatenate(alns)
        gpos = starts[np.clip(aln["chr"], 0, len(lens) - 1)] + aln["pos"] - 1
        for f in first[:6]:
            idx = np.nonzero((aln["chr"] >= 0) & (gpos <= f["pos"]) & (gpos + 400 > f["pos"]))[0][:12]
            near.append({"pos": f["pos"], "reads": [{"read": int(i), **{q: int(aln[q][i]) for q in ("pos", "chr", "flag", "mapq", "nm", "as", "xs", "n_cigar", "fwd")}, "gpos": int(gpos[i])} for i in idx]})
        out["near"] = near
        res["compare"].append(out)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
