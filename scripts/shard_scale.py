"""Eight shards against the single stream at batch sizes of the real thing: a human-like synthetic genome (default 400 Mbp),
2 M pairs x 150 bp as FASTQ files in tmpfs, the native CLI with -sam and -vcf once as one stream and once as eight shards that
share this box's GPU (-devices 0,0,0,0,0,0,0,0), batches of 512 K reads — eight slots of the duplicate-key exchange and of the
SAM place exchange filled with full batches.  SAM and VCF must be the same bytes.
    python scripts/shard_scale.py [--genome-mbp 400] [--batch-pairs 2000000] [--shards 8] [--batch 524288]"""
import argparse, filecmp, json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from mapcaller_amd import api, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=400.0)
    ap.add_argument("--contigs", type=int, default=8)
    ap.add_argument("--batch-pairs", type=int, default=2_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--batch", type=int, default=524288)
    a = ap.parse_args()
    args = argparse.Namespace(genome_mbp=a.genome_mbp, contigs=a.contigs, repeats=2000, genome="human")
    dev = torch.device("cuda", 0)
    codes, lens, _ = bench.make_genome(args, dev, seed=4321)
    tmp = tempfile.mkdtemp(prefix="mcx_scale_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"genome_mbp": a.genome_mbp, "pairs": a.batch_pairs, "shards": a.shards, "batch_reads": a.batch}
    try:
        prefix = os.path.join(tmp, "idx")
        ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=True)
        ix.save(prefix)
        reads = bench.make_reads(codes, lens, a.batch_pairs, 150, seed=99, device=dev).reshape(2 * a.batch_pairs, 150).cpu()
        ix.close()
        del codes
        torch.cuda.empty_cache()
        f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
        synth.write_fastq(f1, reads, 0, 2)
        synth.write_fastq(f2, reads, 1, 2)
        del reads
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mapcaller_amd", "mapcaller-mi355x")
        runs = {}
        for tag, extra in (("single", []), ("shards", ["-devices", ",".join(["0"] * a.shards)])):
            sam, vcf = os.path.join(tmp, tag + ".sam"), os.path.join(tmp, tag + ".vcf")
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-i", prefix, "-f", f1, "-f2", f2, "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-batch", str(a.batch)] + extra,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            runs[tag] = {"seconds": round(time.perf_counter() - t0, 2), "rc": r.returncode, "sam_bytes": os.path.getsize(sam) if os.path.exists(sam) else 0}
            if r.returncode:
                runs[tag]["stderr"] = r.stderr[-600:]
        out["runs"] = runs
        if all(v["rc"] == 0 for v in runs.values()):
            out["sam_identical"] = filecmp.cmp(os.path.join(tmp, "single.sam"), os.path.join(tmp, "shards.sam"), shallow=False)
            body = lambda p: [l for l in open(p, encoding="latin-1") if not l.startswith(("##command_line", "##reference"))]
            va, vb = body(os.path.join(tmp, "single.vcf")), body(os.path.join(tmp, "shards.vcf"))
            out["vcf_identical"] = va == vb
            out["vcf_records"] = sum(1 for l in va if not l.startswith("#"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
