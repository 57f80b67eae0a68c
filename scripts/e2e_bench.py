#!/usr/bin/env python3
"""File-to-file timing of the CLI (FASTQ in, SAM [+VCF] out) next to the in-HBM number of bench.py:
   python scripts/e2e_bench.py [--genome-mbp 200] [--pairs 1000000] [--vcf] [--gz]
Builds a synthetic genome + index (GPU), writes FASTQ files, runs mapcaller-mi355x, reports reads/s
(index load excluded by timing a second run with zero reads)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapcaller_amd import api, synth  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=200.0)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--vcf", action="store_true")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--keep", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(genome_mbp=a.genome_mbp, contigs=8, repeats=200)
    codes, lens = bench.make_genome(args, dev, seed=99)
    index = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=False)
    tmp = a.keep or tempfile.mkdtemp(prefix="mcx_e2e_")
    os.makedirs(tmp, exist_ok=True)
    prefix = os.path.join(tmp, "idx")
    index.save(prefix)
    index.close()
    reads = bench.make_reads(codes, lens, a.pairs, 150, seed=5, device=dev).reshape(2 * a.pairs, 150).cpu()
    del codes
    f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
    synth.write_fastq(f1, reads, 0, 2)
    synth.write_fastq(f2, reads, 1, 2)
    e1, e2 = os.path.join(tmp, "e1.fq"), os.path.join(tmp, "e2.fq")
    open(e1, "w").close(); open(e2, "w").close()
    if a.gz:
        subprocess.run(["gzip", "-1", "-f", f1, f2], check=True)
        f1 += ".gz"; f2 += ".gz"
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam, vcf = os.path.join(tmp, "o.sam"), os.path.join(tmp, "o.vcf")
    tail = ["-alg", "ksw2", "-sam", sam] + (["-vcf", vcf] if a.vcf else ["-no_vcf"])

    def run(x, y):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-i", prefix, "-f", x, "-f2", y] + tail, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, MCX_TIMING="1"))
        for l in r.stderr.split("\n"):
            if l.startswith("[mcx_"):
                print(l, file=sys.stderr)
        return time.perf_counter() - t0
    t_empty = run(e1, e2)
    t_full = run(f1, f2)
    out = {"pairs": a.pairs, "genome_mbp": a.genome_mbp, "gz": a.gz, "vcf": a.vcf, "wall_s": round(t_full, 2), "startup_s": round(t_empty, 2),
           "reads_per_s_file_to_file": round(2 * a.pairs / max(t_full - t_empty, 1e-9), 1), "sam_bytes": os.path.getsize(sam)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
