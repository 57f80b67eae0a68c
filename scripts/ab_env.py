#!/usr/bin/env python3
"""A/B on the GPU box: one genome + index + batches, then the same steps under several environments.

    python scripts/ab_env.py [--steps 3] [--genome human] [--batch-pairs 4000000] "" "MCX_SEED_LANE_PER_BLOCK=1" "A=1 B=2" ...

Every variant gets a fresh Mapper (switches read at context creation take effect); switches that libmcx.so caches in
function-local statics cannot be compared this way.  Prints ms per step and the stage times of each variant."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--genome", default="human")
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--repeats", type=int, default=2000)
    ap.add_argument("--batch-pairs", type=int, default=4_000_000)
    ap.add_argument("--rlen", type=int, default=150)
    ap.add_argument("--alg", default="ksw2")
    ap.add_argument("--sub", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.001)
    ap.add_argument("--dele", type=float, default=0.001)
    ap.add_argument("--full-sa", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=1, help="times the list of variants is gone through")
    ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    from mapcaller_amd import api
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    codes, lens, _ = bench.make_genome(a, dev, seed=1234)
    index = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=a.full_sa)
    n = 2 * a.batch_pairs
    batches = [bench.make_reads(codes, lens, a.batch_pairs, a.rlen, seed=1000 + s, device=dev, sub=a.sub, ins=a.ins, dele=a.dele).reshape(-1).contiguous()
               for s in range(a.steps + 1)]
    del codes
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * a.rlen).to(torch.uint32)
    d_aln = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_cig = torch.empty(api.cigar_pool_words(n), dtype=torch.int32, device=dev)
    for rnd in range(a.rounds):
        for v in a.variants:
            kv = [x.split("=", 1) for x in v.split()]
            for k, val in kv:
                os.environ[k] = val
            mp = api.Mapper(index, alg=a.alg, max_read_len=max(256, a.rlen), max_batch_reads=n)
            mp.map_batch_dev(batches[0].data_ptr(), off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
            before = mp.stats.as_dict()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(1, a.steps + 1):
                mp.map_batch_dev(batches[i].data_ptr(), off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            after = mp.stats.as_dict()
            d = {k: after[k] - before[k] for k in after}
            # what the last step left: a checksum over the records' fields (not the pool offsets: the atomics' order) and every read's CIGAR words
            rec = d_aln.view(torch.int32).reshape(n, 16)
            cig = d_cig
            fields = rec[:, :14].to(torch.int64)
            w = torch.arange(1, 15, device=dev, dtype=torch.int64)
            chk_rec = int(((fields * w).sum(1) * (torch.arange(n, device=dev, dtype=torch.int64) % 1000003 + 1)).sum().item() & 0xFFFFFFFFFFFF)
            n_cig = rec[:, 11].to(torch.int64)
            first = rec[:, 14].to(torch.int64)
            total = int(n_cig.sum().item())
            rid = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), n_cig)
            k_in = torch.arange(total, device=dev, dtype=torch.int64) - torch.repeat_interleave(torch.cumsum(n_cig, 0) - n_cig, n_cig)
            words = cig[first[rid] + k_in].to(torch.int64) & 0xFFFFFFFF
            chk_cig = int(((words * (k_in + 1)) * (rid % 1000003 + 1)).sum().item() & 0xFFFFFFFFFFFF)
            print(json.dumps({"variant": v, "ms_per_step": round(1000 * dt / a.steps, 3), "M_reads_per_s": round(n * a.steps / dt / 1e6, 1),
                              "stage_ms": {k[3:]: round(d[k] / a.steps, 3) for k in d if k.startswith("ms_")}, "tier1_pairs": d["tier1_pairs"],
                              "mapped": d["mapped"], "fm_blocks": d["fm_blocks"], "dp_jobs": d["dp_jobs"], "dp_cells": d["dp_cells"], "simple_pairs": d.get("simple_pairs"),
                              "checksum_records": chk_rec, "checksum_cigars": chk_cig}), flush=True)
            mp.close()
            for k, _ in kv:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
