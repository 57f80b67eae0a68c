#!/usr/bin/env python3
"""The shapes of the gapped-extension problems of a config-5-like sample (250 bp pairs at 2.5 % insertions + 2.5 % deletions per base), taken from the host
emulation of the pipeline (tests/hostemu with MCX_EMU_DUMP_JOBS): how many cells the lane kernels compute for the cells the problems have — the rows of a
wavefront run to its longest query, the columns to a multiple of the strip.  CPU only; test infrastructure."""
import argparse, collections, ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapcaller_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=3000)
ap.add_argument("--rlen", type=int, default=250)
ap.add_argument("--sub", type=float, default=0.005)
ap.add_argument("--ins", type=float, default=0.025)
ap.add_argument("--dele", type=float, default=0.025)
ap.add_argument("--alg", default="nw")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="dpshapes")
g = synth.random_genome([1_500_000], seed=7, n_repeats=10, repeat_len=800, tandem=2, n_runs=2)
fa = os.path.join(tmp, "g.fa"); synth.write_fasta(fa, g)
prefix = os.path.join(tmp, "idx")
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "MapCaller"), "index", fa, prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
bases, _ = synth.simulate_reads(g, a.pairs, a.rlen, True, 11, frag_mean=500, frag_sd=50, frag_min=a.rlen + 24, frag_max=1000, skip_head=3000, skip_contigs=(), sub=a.sub, ins=a.ins, dele=a.dele, n_rate=0.0)
f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
synth.write_fastq(f1, bases, 0, 2); synth.write_fastq(f2, bases, 1, 2)
L = ctypes.CDLL(os.path.join(ROOT, "tests", "hostemu", "libhostemu.so"))
L.hostemu_map_files.restype = ctypes.c_int64
L.hostemu_map_files.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
dump = os.path.join(tmp, "jobs.txt"); os.environ["MCX_EMU_DUMP_JOBS"] = dump
st = (ctypes.c_int64 * 12)()
L.hostemu_map_files(prefix.encode(), f1.encode(), f2.encode(), 0 if a.alg == "nw" else 1, os.path.join(tmp, "e.sam").encode(), 1 << 20, None, 256, st)
jobs = [tuple(map(int, l.split())) for l in open(dump)]
print("problems", len(jobs), "per read", len(jobs) / (2 * a.pairs), "cells", sum(m * n for m, n in jobs))
lists = collections.defaultdict(list)
for m, n in jobs:
    k = "tiny (<=8 x <=8)" if m <= 8 and n <= 8 else "target <=16" if n <= 16 else "target 17-64" if n <= 64 else "target 65-256" if n <= 256 else "larger"
    lists[k].append((m, n))
for k, v in lists.items():
    cells = sum(m * n for m, n in v)
    K = 8 if k.startswith("tiny") else 16
    padc = sum(m * ((n + K - 1) // K * K) for m, n in v)
    # rows to the largest of 128 neighbours of the shape order (strips, rows / 4)
    o = sorted(v, key=lambda x: (-((x[1] + K - 1) // K), -(x[0] >> 2)))
    padrc = 0
    for i in range(0, len(o), 128):
        grp = o[i:i + 128]
        rows = max(m for m, _ in grp); strips = max((n + K - 1) // K for _, n in grp)
        padrc += rows * strips * K * len(grp)
    ms = sorted(m for m, _ in v); ns = sorted(n for _, n in v)
    q = lambda s, f: s[int(f * (len(s) - 1))]
    print(f"{k:18s} problems {len(v):8d} cells {cells:12d} ({cells / max(1, sum(m * n for m, n in jobs)):.3f})  computed/cells: columns {padc / max(1, cells):.2f}, + rows {padrc / max(1, cells):.2f};"
          f" query 10/50/90 % {q(ms, .1)}/{q(ms, .5)}/{q(ms, .9)}  target {q(ns, .1)}/{q(ns, .5)}/{q(ns, .9)}; walk steps / sweep rows {sum(m + n for m, n in v) / max(1, sum(m * ((n + K - 1) // K) for m, n in v)):.2f}")
