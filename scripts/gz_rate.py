"""How fast does the file front end take .gz FASTQ?  1 M pairs x 150 bp as two files in tmpfs — plain, gzip, and bgzip's container
(independent members the reader inflates side by side) — mapped without SAM output.
    python scripts/gz_rate.py"""
import json, os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import bench
from mapcaller_amd import api, synth


def write_bgzf(path, data, block=0xff00, level=1):
    import struct, zlib
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            chunk = data[i:i + block]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs in the two files (the default run is a quarter of a second: mostly the pipeline filling; 4 M pairs show the rate)")
    ap.add_argument("--only-gz", type=int, default=0, help="1: skip the plain and bgzip runs")
    a = ap.parse_args()
    args = argparse.Namespace(genome_mbp=100.0, contigs=4, repeats=200, genome="uniform")
    dev = torch.device("cuda", 0)
    codes, lens, _ = bench.make_genome(args, dev, seed=5)
    ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=True)
    n_pairs = a.pairs
    reads = bench.make_reads(codes, lens, n_pairs, 150, seed=9, device=dev).reshape(2 * n_pairs, 150).cpu()
    tmp = tempfile.mkdtemp(prefix="mcx_gz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
        synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
        mp = api.Mapper(ix, alg="ksw2", max_read_len=256, max_batch_reads=1 << 20)
        out = {}
        p1, p2 = f1, f2
        for tag in (("gz",) if a.only_gz else ("plain", "bgzf", "gz")):
            if tag == "bgzf":
                f1, f2 = os.path.join(tmp, "b1.fq.gz"), os.path.join(tmp, "b2.fq.gz")
                write_bgzf(f1, open(p1, "rb").read()); write_bgzf(f2, open(p2, "rb").read())
            if tag == "gz":
                procs = [subprocess.Popen(["gzip", "-6", f]) for f in (p1, p2)]  # (the two files side by side)
                assert all(q.wait() == 0 for q in procs)
                f1, f2 = p1 + ".gz", p2 + ".gz"
            for variant in (("gz", "gz_zlib_one_thread") if tag == "gz" else (tag,)):
                if variant == "gz_zlib_one_thread":
                    os.environ["MCX_GZ_SERIAL"] = "1"  # (the reader of rounds 1-5: zlib's gzread on a thread of its own per file)
                try:
                    mp.reset(); mp.map_files(f1, f2, None)
                    mp.reset()
                    t = time.perf_counter()
                    st = mp.map_files(f1, f2, None)
                    dt = time.perf_counter() - t
                finally:
                    os.environ.pop("MCX_GZ_SERIAL", None)
                out[variant] = {"reads_per_s": round(st["reads"] / dt), "seconds": round(dt, 3), "bytes": os.path.getsize(f1) + os.path.getsize(f2), "reads": st["reads"]}
            if tag == "gz":  # the reader by itself: text per second
                import ctypes as C
                L = api.lib()
                L.mcx_gz_inflate.restype = C.c_int64
                L.mcx_gz_inflate.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
                for th in (1, 2, 4, 6, 8, 12, 16):
                    t = time.perf_counter()
                    n = L.mcx_gz_inflate(f1.encode(), th, 2 << 20, None, 0, None)
                    dt = time.perf_counter() - t
                    out.setdefault("inflate_alone_mb_per_s", {})[str(th)] = round(n / dt / 1e6)
        mp.close()
        print(json.dumps(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
