#!/usr/bin/env python3
"""Which pairs the straight-line path (mcx_simple.h) takes, and why the others leave it — on the CPU, through tests/hostemu (the device
headers compiled for the host), on a small genome of bench.py's human-like kind.  Also the A/B that the path changes nothing: the SAM with
it, without its DP problems and with every pair on the general path must be the same file.

    python scripts/simple_path_census.py [--mbp 24] [--pairs 200000] [--alg ksw2] [--sub 0.005 --ins 0.001 --dele 0.001]

Needs the compiled reference for the index (oracle/_ref/MapCaller: `make -C oracle ref`, this container only)."""
import argparse
import ctypes
import filecmp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mapcaller_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=24.0)
    ap.add_argument("--pairs", type=int, default=200000)
    ap.add_argument("--rlen", type=int, default=150)
    ap.add_argument("--alg", default="ksw2")
    ap.add_argument("--sub", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.001)
    ap.add_argument("--dele", type=float, default=0.001)
    a = ap.parse_args()
    dev = torch.device("cpu")
    g = argparse.Namespace(genome_mbp=a.mbp, contigs=4, repeats=max(4, int(2000 * a.mbp / 3100)), genome="human")
    codes, lens, note = bench.make_genome(g, dev, seed=1234)
    d = tempfile.mkdtemp(prefix="census_")
    fa = os.path.join(d, "g.fa")
    off = 0
    with open(fa, "w") as f:
        for i, n in enumerate(lens):
            seq = bytes(b"ACGT"[c] for c in codes[off:off + n].numpy()).decode()
            off += n
            f.write(f">chr{i + 1}\n")
            f.writelines(seq[k:k + 80] + "\n" for k in range(0, n, 80))
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "MapCaller"), "index", fa, os.path.join(d, "idx")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = bench.make_reads(codes, lens, a.pairs, a.rlen, seed=1001, device=dev, sub=a.sub, ins=a.ins, dele=a.dele).reshape(2 * a.pairs, a.rlen)
    f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
    synth.write_fastq(f1, reads, 0, 2)
    synth.write_fastq(f2, reads, 1, 2)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostemu")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
    lib.hostemu_map_files.restype = ctypes.c_int64
    lib.hostemu_map_files.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int64)]
    out = {}
    for tag, env in (("path", {"MCX_EMU_SIMPLE_WHY": "1"}), ("no_dp", {"MCX_EMU_SIMPLE_NO_DP": "1"}), ("general", {"MCX_EMU_NO_SIMPLE": "1"})):
        for k in ("MCX_EMU_SIMPLE_WHY", "MCX_EMU_SIMPLE_NO_DP", "MCX_EMU_NO_SIMPLE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        st = (ctypes.c_int64 * 12)()
        sam = os.path.join(d, tag + ".sam")
        lib.hostemu_map_files(os.path.join(d, "idx").encode(), f1.encode(), f2.encode(), 0 if a.alg == "nw" else 1, sam.encode(), 1 << 20, None, 256, st)
        out[tag] = (sam, list(st))
        print(f"{tag:8s} straight-line pairs {st[11]:8d} of {a.pairs}, DP problems left to the general path {st[6]:8d}", flush=True)
    same = filecmp.cmp(out["path"][0], out["general"][0], shallow=False) and filecmp.cmp(out["no_dp"][0], out["general"][0], shallow=False)
    print("SAM identical three ways" if same else "SAM DIFFERS")
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
