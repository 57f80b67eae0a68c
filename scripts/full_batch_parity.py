#!/usr/bin/env python3
"""A FULL bench batch against the compiled reference ("full-size parity is a prefix").

bench.py times batches of 4 M pairs (8 M reads); the suite compares 60 k-pair prefixes with the reference.  This script maps
one whole bench batch of BASELINE config 3 (4 M pairs x 150 bp, -alg ksw2 — the first timed batch of `bench.py` rank 0) as ONE
batch of 8 M reads through the product's file path, and 1 M pairs of config 5 (250 bp, 5 % indels per base, -alg nw) as one
batch of 2 M reads, and compares the SAM files line for line with `oracle/_ref/MapCaller -t 1` (the real reference, built by
oracle/Makefile; -t 1 because its insert-size feedback depends on thread timing otherwise).  The reference runs of both
configs start in the background (one host core each, ~7-8 minutes at ~19 k / ~5 k reads/s) while the GPU maps.

Everything happens in one process: bench.py's synthetic genome is built here, indexed on the GPU by the product's builder,
and the same index files are what the reference loads.  Result: one JSON object (stdout and --out).

  python scripts/full_batch_parity.py --out gpurun_out/full_batch_parity.json
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def file_md5(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def differing_lines(a, b, limit=3):
    """Lines that differ between two SAM files (+ a few examples), streamed."""
    n, ex, na, nb = 0, [], 0, 0
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            la, lb = fa.readline(), fb.readline()
            if not la and not lb:
                break
            na += bool(la)
            nb += bool(lb)
            if la != lb:
                n += 1
                if len(ex) < limit:
                    ex.append((la[:300].decode("latin-1"), lb[:300].decode("latin-1")))
    return n, ex, na, nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs3", type=int, default=4_000_000, help="pairs of the config-3 batch (bench.py's --batch-pairs); 0 = skip")
    ap.add_argument("--pairs5", type=int, default=1_000_000, help="pairs of the config-5 batch; 0 = skip")
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import bench
    from mapcaller_amd import api, synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    if not os.path.exists(ref_bin):
        sys.exit("oracle/_ref/MapCaller is not on this box (make -C oracle ref where /root/reference exists)")
    dev = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp(prefix="mcx_fbp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    res = {"genome_mbp": a.genome_mbp, "checker": "oracle/_ref/MapCaller -t 1 (the reference, compiled from its own sources)", "configs": []}
    try:
        gargs = argparse.Namespace(genome_mbp=a.genome_mbp, contigs=a.contigs, repeats=2000, genome="human")
        codes, lens, note = bench.make_genome(gargs, dev, seed=1234)
        t0 = time.perf_counter()
        ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=2)
        res["index_build_s"] = round(time.perf_counter() - t0, 2)
        prefix = os.path.join(tmp, "idx")
        ix.save(prefix)
        jobs = []
        if a.pairs3:
            jobs.append(dict(name="config 3: 150 bp PE, -alg ksw2, one batch of the bench", pairs=a.pairs3, rlen=150, alg="ksw2", ins=0.001, dele=0.001, tag="c3"))
        if a.pairs5:
            jobs.append(dict(name="config 5: 250 bp PE at 5 % indels per base, -alg nw", pairs=a.pairs5, rlen=250, alg="nw", ins=0.025, dele=0.025, tag="c5"))
        # reads and the reference runs first (they take minutes on one core each), the GPU maps meanwhile
        for j in jobs:
            reads = bench.make_reads(codes, lens, j["pairs"], j["rlen"], seed=1001, device=dev, sub=0.005, ins=j["ins"], dele=j["dele"])
            reads = reads.reshape(2 * j["pairs"], j["rlen"]).cpu()
            j["f1"], j["f2"] = os.path.join(tmp, j["tag"] + "_1.fq"), os.path.join(tmp, j["tag"] + "_2.fq")
            synth.write_fastq(j["f1"], reads, 0, 2)
            synth.write_fastq(j["f2"], reads, 1, 2)
            del reads
            j["chk"] = os.path.join(tmp, j["tag"] + "_ref.sam")
            cmd = [ref_bin, "-i", prefix, "-f", j["f1"], "-f2", j["f2"], "-alg", j["alg"], "-sam", j["chk"], "-no_vcf", "-t", "1", "-log", os.path.join(tmp, j["tag"] + ".log")]
            j["t_ref"] = time.perf_counter()
            j["proc"] = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        del codes
        torch.cuda.empty_cache()
        for j in jobs:
            n_reads = 2 * j["pairs"]
            mp = api.Mapper(ix, alg=j["alg"], max_read_len=256, max_batch_reads=n_reads)  # the whole file is ONE batch
            j["out"] = os.path.join(tmp, j["tag"] + "_gpu.sam")
            t0 = time.perf_counter()
            st = mp.map_files(j["f1"], j["f2"], j["out"])
            j["gpu_s"] = time.perf_counter() - t0
            j["st"] = st
            mp.close()
        for j in jobs:
            err = j["proc"].communicate()[1]
            j["ref_s"] = time.perf_counter() - j["t_ref"]
            if j["proc"].returncode != 0:
                raise RuntimeError("the reference failed: " + err[-400:])
            nd, ex, na, nb = differing_lines(j["chk"], j["out"])
            st = j["st"]
            res["configs"].append({
                "config": j["name"], "pairs": j["pairs"], "reads_in_one_batch": 2 * j["pairs"], "differing_lines": nd, "examples": ex,
                "sam_lines": {"reference": na, "gpu": nb}, "sam_bytes": os.path.getsize(j["out"]),
                "sam_md5": {"reference": file_md5(j["chk"]), "gpu": file_md5(j["out"])},
                "mapped": st["mapped"], "tier1_pairs": st["tier1_pairs"], "dp_jobs": st["dp_jobs"], "replayed_pairs": st["replayed_pairs"],
                "halved_selections": st["halved_selections"],
                "gpu_file_to_file_s": round(j["gpu_s"], 2), "reference_t1_wall_s": round(j["ref_s"], 1),
                "reference_t1_reads_per_s_incl_index_load": round(2 * j["pairs"] / j["ref_s"], 1)})
        res["genome"] = note
        res["identical"] = all(c["differing_lines"] == 0 for c in res["configs"])
        ix.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    line = json.dumps(res)
    print(line, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(json.dumps(res, indent=1) + "\n")
    sys.exit(0 if res["identical"] else 1)


if __name__ == "__main__":
    main()
