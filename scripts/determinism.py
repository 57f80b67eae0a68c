"""Maps one bench batch several times on the GPU and compares the records run against run (every field, every CIGAR).
    python scripts/determinism.py [--runs 3] [bench.py's workload flags]
Prints one JSON line: differing reads per run pair and a few examples."""
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
    paired = not args.single_end
    n = (2 if paired else 1) * args.batch_pairs
    batch = bench.make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1001, device=dev, sub=args.sub, ins=args.ins, dele=args.dele, paired=paired).reshape(-1).contiguous()
    del codes
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    mapper = api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=n)
    outs = []
    for r in range(runs):
        d_aln = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        d_cig = torch.zeros(api.cigar_pool_words(n), dtype=torch.int32, device=dev)
        mapper.reset()
        mapper.map_batch_dev(batch.data_ptr(), off.data_ptr(), n, paired, d_aln.data_ptr(), d_cig.data_ptr())
        torch.cuda.synchronize()
        aln = d_aln.cpu().numpy().view(api.ALN_DTYPE)
        cig = d_cig.cpu().numpy().view(np.uint32)
        outs.append((aln, cig))
    fields = [f for f in api.ALN_DTYPE.names if f not in ("cigar_off", "pad")]
    res = {"reads": n, "runs": runs, "pairs": []}
    a0, c0 = outs[0]
    for r in range(1, runs):
        a1, c1 = outs[r]
        bad = np.zeros(n, dtype=bool)
        per_field = {}
        for f in fields:
            m = a0[f] != a1[f]
            if m.any():
                per_field[f] = int(m.sum())
            bad |= m
        # CIGARs of reads whose counts agree: word by word
        same_n = ~bad
        nc = a0["n_cigar"].astype(np.int64)
        idx = np.nonzero(same_n & (nc > 0))[0]
        o0 = a0["cigar_off"][idx].astype(np.int64); o1 = a1["cigar_off"][idx].astype(np.int64); k = nc[idx]
        rep = np.repeat(np.arange(idx.size), k)
        within = np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)
        dif = c0[o0[rep] + within] != c1[o1[rep] + within]
        cig_bad = np.zeros(idx.size, dtype=bool)
        np.logical_or.at(cig_bad, rep[dif], True)
        if cig_bad.any():
            per_field["cigar"] = int(cig_bad.sum())
            bad[idx[cig_bad]] = True
        ex = []
        for i in np.nonzero(bad)[0][:6]:
            ex.append({"read": int(i), "run0": {f: int(a0[f][i]) for f in fields}, f"run{r}": {f: int(a1[f][i]) for f in fields}})
        res["pairs"].append({"run": r, "differing_reads": int(bad.sum()), "per_field": per_field, "examples": ex})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
