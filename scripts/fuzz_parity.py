#!/usr/bin/env python3
"""Randomised differential check of the product (mapcaller-mi355x on the GPU) against the CPU oracle.

    python scripts/fuzz_parity.py --rounds 30 [--seed 1] [--keep DIR]

Every round draws a small genome (contigs, repeats, tandem runs, N runs), a donor with SNPs/indels,
read length, insert size, error rates, single or paired end, FASTQ or FASTA, algorithm — maps the
reads with both and compares the SAM line by line and the VCF body.  tests/test_gpu_parity.py runs 60 + 25
rounds with fixed seeds; longer runs by hand.  A divergence prints the round's parameters and keeps its files.
Checker use only: the oracle is the thing compared against, never a fallback.
"""
import argparse
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapcaller_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
ORACLE = os.path.join(ROOT, "oracle", "mcx_oracle")
REF = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
MODE = "gpu"  # "gpu": product vs oracle (needs a GPU); "ref": oracle vs the compiled reference (CPU only: pins the oracle)
NO_VCF = False  # --no-vcf: mapping alone, the SAM compared
CLI_ARGS = []  # extra switches for the product's command line (e.g. -devices 0,0,0 -batch 400: the reads dealt to three shards)


def body(path, vcf=False):
    lines = open(path, "rb").read().decode("latin-1").split("\n")  # (binary: the reference's stray quality byte may be a carriage return)
    if vcf:
        return [l for l in lines if not l.startswith(("##command_line=", "##reference="))]
    return lines


def mask_se_reverse_qual(lines):
    out = []
    for l in lines:
        f = l.split("\t")
        if len(f) > 10 and f[1].isdigit() and (int(f[1]) & 0x11) == 0x10:
            f[10] = "?"
        out.append("\t".join(f))
    return out


def sam_bytes_differ(ref_path, ora_path):
    """Records of the reference's SAM that differ from the oracle's.  The reference leaves the first
    QUAL byte of a reverse-strand single-end FASTQ record uninitialised: it can be any
    byte — tab, newline, or NUL, which ends the printed string there and leaves QUAL empty — so lines
    cannot be paired up.  Walk the oracle's records and match the reference's bytes against each,
    allowing exactly that freedom on exactly those records."""
    a, b = open(ref_path, "rb").read(), open(ora_path, "rb").read()
    if a == b:
        return 0
    pos, bad = 0, 0
    for line in b.split(b"\n")[:-1]:
        line += b"\n"
        if a.startswith(line, pos):
            pos += len(line)
            continue
        f = line.split(b"\t")
        if len(f) > 10 and f[1].isdigit() and (int(f[1]) & 0x11) == 0x10:
            head = sum(len(x) + 1 for x in f[:10])
            tail, after = line[head + 1:], line[head + len(f[10]):]
            if a.startswith(line[:head], pos):
                if a.startswith(tail, pos + head + 1):
                    pos += len(line)
                    continue
                if a.startswith(after, pos + head):
                    pos += head + len(after)
                    continue
        bad += 1
        nxt = a.find(b"\n" + line[:line.find(b"\t") + 1], pos)  # resynchronise on the next read name
        pos = a.find(b"\n", max(pos, nxt) + 1) + 1 if nxt >= 0 else pos
    return bad + (1 if pos != len(a) else 0)


WIDE = False  # --wide: longer reads and more of the variant-calling switches (drawn after the usual ones: the usual rounds keep their numbers)


def draw(rng):
    """The parameters of one round (every random draw happens here, so a round can be replayed by number)."""
    d = {}
    d["lens"] = [rng.randint(40_000, 400_000) for _ in range(rng.randint(1, 4))]
    d["genome"] = dict(seed=rng.randint(1, 1 << 30), n_repeats=rng.randint(0, 30), repeat_len=rng.choice([300, 800, 2000]),
                       tandem=rng.randint(0, 10), n_runs=rng.randint(0, 6))
    d["donor"] = dict(seed=rng.randint(1, 1 << 30), snp=rng.choice([0.0, 0.002, 0.01]), indel=rng.choice([0.0, 0.0005, 0.002]))
    d["rlen"] = rng.choice([36, 75, 100, 150, 151, 250, 300])
    d["paired"] = rng.random() < 0.7
    d["fastq"] = rng.random() < 0.8 or not d["paired"]
    d["frag_mean"] = rng.choice([250, 400, 500, 800]) + d["rlen"]
    d["n"] = rng.randint(300, 4000)
    d["err"] = dict(sub=rng.choice([0.0, 0.005, 0.02, 0.05]), ins=rng.choice([0.0, 0.001, 0.01]), dele=rng.choice([0.0, 0.001, 0.01]),
                    n_rate=rng.choice([0.0, 0.0, 0.003]))
    d["reads_seed"] = rng.randint(1, 1 << 30)
    d["frag_sd"] = rng.choice([10, 50, 120])
    d["alg"] = rng.choice(["nw", "ksw2"])
    d["vcf"] = rng.choice([[], ["-gvcf"], ["-filter"], ["-ploidy", "1"], ["-somatic"], ["-ad", "3", "-min_gap", "20"]])
    if WIDE:
        if rng.random() < 0.35:
            d["rlen"] = rng.choice([400, 600, 900])
            d["frag_mean"] = rng.choice([300, 600]) + d["rlen"]
            d["n"] = min(d["n"], 1200)
        extra = rng.choice([[], ["-monomorphic"], ["-dup", "2"], ["-maxclip", "12"], ["-min_cnv", "20"], ["-size", "300"], ["-gvcf", "-filter"]])
        if not any(x in d["vcf"] for x in extra if x.startswith("-")):
            d["vcf"] = d["vcf"] + extra
    return d


def one_round(d, tmp):
    lens, rlen, paired, fastq, mean, n, p = d["lens"], d["rlen"], d["paired"], d["fastq"], d["frag_mean"], d["n"], d["err"]
    gp = dict(d["genome"]); gseed = gp.pop("seed")
    g = synth.random_genome(lens, seed=gseed, **gp)
    fa = os.path.join(tmp, "g.fa")
    synth.write_fasta(fa, g)
    prefix = os.path.join(tmp, "idx")
    subprocess.run([REF if MODE == "ref" else EXE, "index", fa, prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    donor = synth.mutate_genome(g, d["donor"]["seed"], snp=d["donor"]["snp"], indel=d["donor"]["indel"])
    bases, _ = synth.simulate_reads(donor, n, rlen, paired, d["reads_seed"], frag_mean=mean, frag_sd=d["frag_sd"],
                                    frag_min=rlen + 24, frag_max=mean + 500, skip_head=3000, skip_contigs=(0,), **p)
    ext = "fq" if fastq else "fa"
    f1, f2 = os.path.join(tmp, f"r1.{ext}"), os.path.join(tmp, f"r2.{ext}")
    writer = synth.write_fastq if fastq else synth.write_fasta_reads
    if paired:
        writer(f1, bases, 0, 2); writer(f2, bases, 1, 2)
        files = ["-f", f1, "-f2", f2]
    else:
        writer(f1, bases, 0, 1)
        files = ["-f", f1]
    alg, vcf_flags = d["alg"], d["vcf"]
    desc = dict(lens=lens, rlen=rlen, paired=paired, fastq=fastq, n=n, alg=alg, vcf=vcf_flags, frag_mean=mean, two_base=bool(d.get("two_base")), **p)
    gs, gv, os_, ov = (os.path.join(tmp, x) for x in ("gpu.sam", "gpu.vcf", "ora.sam", "ora.vcf"))
    if MODE == "ref":
        subprocess.run([REF, "-i", prefix, *files, "-alg", alg, "-sam", gs, "-vcf", gv, "-t", "1", "-log", os.path.join(tmp, "job.log"), *vcf_flags],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        walk = ["-two_base"] if d.get("two_base") else []  # every other round: the seeding walk over the pair records
        if NO_VCF:  # mapping alone: no alignment profile is kept, so the straight-line path (k_simple) is open to the pairs (with MCX_ORDER_MIN=1 on these small batches)
            subprocess.run([EXE, "-i", prefix, *files, "-alg", alg, "-sam", gs, "-no_vcf", *walk, *CLI_ARGS], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        else:
            subprocess.run([EXE, "-i", prefix, *files, "-alg", alg, "-sam", gs, "-vcf", gv, *vcf_flags, *walk, *CLI_ARGS], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([ORACLE, "-i", prefix, *files, "-alg", alg, "-sam", os_], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([ORACLE, "-i", prefix, *files, "-alg", alg, "-vcf", ov, *vcf_flags], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if MODE == "ref":
        sam_bad = sam_bytes_differ(gs, os_)
    else:
        a, b = body(gs), body(os_)
        sam_bad = sum(1 for x, y in zip(a, b) if x != y) + abs(len(a) - len(b))
    if NO_VCF and MODE != "ref":
        return desc, sam_bad, 0
    va, vb = body(gv, True), body(ov, True)
    vcf_bad = sum(1 for x, y in zip(va, vb) if x != y) + abs(len(va) - len(vb))
    return desc, sam_bad, vcf_bad


def keep(tmp, to, d, a):
    """A round that differed: its files (genome, index, reads, both SAMs / VCFs) and how it was run, where the next look can find them."""
    try:
        os.makedirs(os.path.dirname(to), exist_ok=True)
        shutil.copytree(tmp, to, dirs_exist_ok=True)
        with open(os.path.join(to, "HOW.json"), "w") as fh:
            json.dump({"draw": d, "argv": sys.argv, "cli_args": CLI_ARGS, "no_vcf": NO_VCF, "wide": WIDE, "mode": MODE,
                       "env": {k: v for k, v in os.environ.items() if k.startswith("MCX_")}}, fh, default=str, indent=1)
        print(f"  kept in {to}", flush=True)
    except OSError as e:
        print(f"  (could not keep the round: {e})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keep", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_fail"),
                    help="where a round that differs (or fails) leaves its inputs and both outputs — ON by default since round 6: round 5 lost its one differing round "
                         "(gpurun_out/ comes back from the GPU box); '' = keep nothing")
    ap.add_argument("--only", type=int, default=-1, help="replay just this round of the sequence")
    ap.add_argument("--ref", action="store_true", help="compare the oracle with the compiled reference (oracle/_ref, CPU only) instead of the GPU product")
    ap.add_argument("--cli-args", default="", help="extra switches for mapcaller-mi355x, space separated")
    ap.add_argument("--no-vcf", action="store_true", help="map with -no_vcf and compare the SAM only (opens the straight-line path to the pairs)")
    ap.add_argument("--wide", action="store_true", help="longer reads (up to 900 bp) and more variant-calling switches")
    a = ap.parse_args()
    global MODE, CLI_ARGS, WIDE, NO_VCF
    WIDE = a.wide
    MODE = "ref" if a.ref else "gpu"
    CLI_ARGS = a.cli_args.split()
    NO_VCF = a.no_vcf
    rng = random.Random(a.seed)
    bad = ran = 0
    for r in range(a.rounds):
        d = draw(rng)
        d["two_base"] = r % 2 == 1
        if a.only >= 0 and r != a.only:
            continue
        ran += 1
        tmp = tempfile.mkdtemp(prefix=f"fuzz{r}_")
        keep_to = os.path.join(a.keep, f"seed{a.seed}_round{r}") if a.keep else ""
        try:
            desc, sam_bad, vcf_bad = one_round(d, tmp)
        except subprocess.CalledProcessError as e:
            print(f"round {r}: command failed (exit {e.returncode}): {e.cmd[:6]} ... {d}", flush=True)
            bad += 1
            if keep_to:
                keep(tmp, keep_to, d, a)
            shutil.rmtree(tmp, ignore_errors=True)
            continue
        status = "ok" if not (sam_bad or vcf_bad) else f"DIFF sam={sam_bad} vcf={vcf_bad}"
        print(f"round {r}: {status} {desc}", flush=True)
        if sam_bad or vcf_bad:
            bad += 1
            if keep_to:
                keep(tmp, keep_to, d, a)
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"{ran - bad} of {ran} rounds identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
