"""How fast does the file front end take .gz FASTQ?  1 M pairs x 150 bp as two gzip files in tmpfs, mapped without SAM output.
    python scripts/gz_rate.py"""
import json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import bench
from mapcaller_amd import api, synth


def main():
    args = argparse.Namespace(genome_mbp=100.0, contigs=4, repeats=200, genome="uniform")
    dev = torch.device("cuda", 0)
    codes, lens, _ = bench.make_genome(args, dev, seed=5)
    ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=True)
    n_pairs = 1_000_000
    reads = bench.make_reads(codes, lens, n_pairs, 150, seed=9, device=dev).reshape(2 * n_pairs, 150).cpu()
    tmp = tempfile.mkdtemp(prefix="mcx_gz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
        synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
        mp = api.Mapper(ix, alg="ksw2", max_read_len=256, max_batch_reads=1 << 20)
        out = {}
        for tag in ("plain", "gz"):
            if tag == "gz":
                subprocess.run(["gzip", "-1", f1, f2], check=True)
                f1, f2 = f1 + ".gz", f2 + ".gz"
            mp.reset(); mp.map_files(f1, f2, None)
            mp.reset()
            t = time.perf_counter()
            st = mp.map_files(f1, f2, None)
            dt = time.perf_counter() - t
            out[tag] = {"reads_per_s": round(st["reads"] / dt), "seconds": round(dt, 3), "bytes": os.path.getsize(f1) + os.path.getsize(f2)}
        mp.close()
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
