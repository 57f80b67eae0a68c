"""Experiment: two mapping contexts on one GPU, a host thread each, alternate batches — what does the GPU gain when the
VALU-bound and the request-bound kernels of two batches overlap?   python scripts/two_ctx.py [bench flags]"""
import json, os, sys, threading, time
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
    K = 6
    batches = [bench.make_reads(codes, lens, args.batch_pairs, args.rlen, seed=1000 + s, device=dev, sub=args.sub, ins=args.ins, dele=args.dele).reshape(-1).contiguous() for s in range(K)]
    del codes
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * args.rlen).to(torch.uint32)
    res = {}
    for n_ctx in (1, 2):
        ctx = []
        for c in range(n_ctx):
            ctx.append((api.Mapper(index, alg=args.alg, max_read_len=max(256, args.rlen), max_batch_reads=n),
                        torch.empty(n * 64, dtype=torch.uint8, device=dev), torch.empty(api.cigar_pool_words(n), dtype=torch.int32, device=dev)))

        def work(c, rounds):
            mp, aln, cig = ctx[c]
            for r in range(rounds):
                for k in range(c, K, n_ctx):
                    mp.map_batch_dev(batches[k].data_ptr(), off.data_ptr(), n, True, aln.data_ptr(), cig.data_ptr())

        def run(rounds):
            th = [threading.Thread(target=work, args=(c, rounds)) for c in range(n_ctx)]
            torch.cuda.synchronize()
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t

        run(1)
        dt = run(2)
        res[f"{n_ctx}_contexts_ms_per_batch"] = round(1000 * dt / (2 * K), 3)
        for mp, _, _ in ctx:
            mp.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
