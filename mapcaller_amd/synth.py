"""Synthetic genomes and reads for tests and bench (the reference ships no read
simulator; its sv_simulator only mutates genomes).

Everything is vectorised torch so that the same code makes a few thousand pairs on the CPU for
the parity tests and tens of millions directly in HBM for ``bench.py`` (the bench's timed region
starts with the reads already resident on the device).

Conventions
-----------
* base codes: A,C,G,T = 0..3 and N = 4 (``nst_nt4_table``, reference src/BWT_Index/bntseq.c:40).
* a read batch is a dense ``uint8 [n_reads, rlen]`` tensor of ASCII bases; for paired-end data
  mates are interleaved (rows 2p and 2p+1), which is how the reference lays out ``ReadArr``
  (src/GetData.cpp:85-99).
* mate 2 is the reverse complement of the far end of the fragment (Illumina FR), so after the
  reference's own ``ReverseOrientation`` (src/ReadMapping.cpp:451) both mates hit one strand.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np
import torch

ASCII = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8)  # A C G T N


@dataclass
class Genome:
    names: List[str]
    codes: List[torch.Tensor]  # uint8 codes 0..4 per contig (CPU or device)

    @property
    def total_len(self) -> int:
        return int(sum(int(c.numel()) for c in self.codes))


def random_genome(lengths: Sequence[int], seed: int, n_repeats: int = 0, repeat_len: int = 2000,
                  tandem: int = 0, n_runs: int = 0, device: str = "cpu") -> Genome:
    """Uniform random contigs with optional planted dispersed repeats (each repeat unit copied to
    two places, 1 % diverged), tandem repeats (a 20-60 bp unit repeated ~12x) and runs of N."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    names, codes = [], []
    for ci, L in enumerate(lengths):
        c = torch.randint(0, 4, (L,), generator=g, dtype=torch.uint8)
        names.append(f"chr{ci + 1}")
        codes.append(c)
    total = sum(lengths)
    for _ in range(n_repeats):
        rl = min(repeat_len, min(lengths) // 4)
        unit = torch.randint(0, 4, (rl,), generator=g, dtype=torch.uint8)
        for _copy in range(2):
            ci = int(torch.randint(0, len(lengths), (1,), generator=g))
            pos = int(torch.randint(0, lengths[ci] - rl, (1,), generator=g))
            u = unit.clone()
            if _copy == 1:
                m = torch.rand(rl, generator=g) < 0.01
                u[m] = (u[m] + 1 + torch.randint(0, 3, (int(m.sum()),), generator=g, dtype=torch.uint8)) % 4
                if int(torch.randint(0, 2, (1,), generator=g)):
                    u = (3 - u).flip(0)
            codes[ci][pos:pos + rl] = u
    for _ in range(tandem):
        ci = int(torch.randint(0, len(lengths), (1,), generator=g))
        ul = int(torch.randint(20, 60, (1,), generator=g))
        reps = 12
        if lengths[ci] <= ul * reps + 10:
            continue
        pos = int(torch.randint(0, lengths[ci] - ul * reps, (1,), generator=g))
        unit = torch.randint(0, 4, (ul,), generator=g, dtype=torch.uint8)
        codes[ci][pos:pos + ul * reps] = unit.repeat(reps)
    for _ in range(n_runs):
        ci = int(torch.randint(0, len(lengths), (1,), generator=g))
        rl = int(torch.randint(1, 40, (1,), generator=g))
        pos = int(torch.randint(0, lengths[ci] - rl, (1,), generator=g))
        codes[ci][pos:pos + rl] = 4
    del total
    return Genome(names, [c.to(device) for c in codes])


def read_fasta(path: str) -> Genome:
    names, seqs, cur = [], [], []
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                if names:
                    seqs.append("".join(cur))
                names.append(line[1:].split()[0])
                cur = []
            else:
                cur.append(line.strip())
    seqs.append("".join(cur))
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ord(ch)] = v
    codes = [torch.from_numpy(lut[np.frombuffer(s.encode(), dtype=np.uint8)].copy()) for s in seqs]
    return Genome(names, codes)


def write_fasta(path: str, genome: Genome, width: int = 70) -> None:
    with open(path, "w") as fh:
        for name, c in zip(genome.names, genome.codes):
            s = ASCII[c.cpu().long()].numpy().tobytes().decode()
            fh.write(f">{name}\n")
            for i in range(0, len(s), width):
                fh.write(s[i:i + width] + "\n")


def mutate_genome(genome: Genome, seed: int, snp: float = 0.002, indel: float = 0.0002,
                  max_indel: int = 8) -> Genome:
    """A donor genome: SNPs plus short insertions/deletions (for the -vcf rows)."""
    rng = np.random.default_rng(seed)
    out = []
    for c in genome.codes:
        a = c.cpu().numpy()
        pieces, i, L = [], 0, len(a)
        n_ev = rng.poisson(L * indel)
        pos = np.sort(rng.integers(10, max(11, L - 10), n_ev))
        for p in pos:
            if p <= i:
                continue
            pieces.append(a[i:p])
            k = int(rng.integers(1, max_indel + 1))
            if rng.random() < 0.5:
                pieces.append(rng.integers(0, 4, k).astype(np.uint8))
                i = p
            else:
                i = min(L, p + k)
        pieces.append(a[i:])
        b = np.concatenate(pieces)
        m = (rng.random(len(b)) < snp) & (b < 4)
        b[m] = (b[m] + rng.integers(1, 4, int(m.sum()))) % 4
        out.append(torch.from_numpy(b.astype(np.uint8)))
    return Genome([n for n in genome.names], out)


def structural_donor(genome: Genome, seed: int, inv: Tuple[int, int, int] = (0, 20000, 3000),
                     move: Tuple[int, int, int, int] = (1, 10000, 3000, 30000)) -> Genome:
    """Two haplotypes of a donor (contigs doubled: all of haplotype A, then all of haplotype B) that
    share homozygous SNPs / short indels, differ by heterozygous ones, and both carry one inversion
    ``inv`` = (contig, start, length) and one moved segment ``move`` = (contig, start, length, new
    start) — the events the -vcf path reports as <INV> / <TNL> break points."""
    hap_a = mutate_genome(genome, seed, snp=0.003, indel=0.0006, max_indel=5)
    hap_b = mutate_genome(hap_a, seed + 1, snp=0.002, indel=0.0003, max_indel=4)
    names, codes = [], []
    for tag, hap in (("a", hap_a), ("b", hap_b)):
        for ci, c in enumerate(hap.codes):
            a = c.clone()
            if ci == inv[0]:
                s0, ln = inv[1], inv[2]
                seg = a[s0:s0 + ln].flip(0)
                a[s0:s0 + ln] = torch.where(seg < 4, 3 - seg, seg)
            if ci == move[0]:
                s0, ln, to = move[1], move[2], move[3]
                seg = a[s0:s0 + ln].clone()
                rest = torch.cat([a[:s0], a[s0 + ln:]])
                to2 = to - ln if to > s0 else to
                a = torch.cat([rest[:to2], seg, rest[to2:]])
            names.append(f"{hap.names[ci]}_{tag}")
            codes.append(a)
    return Genome(names, codes)


def _apply_errors(src: torch.Tensor, rlen: int, sub: float, ins: float, dele: float, n_rate: float,
                  g: torch.Generator) -> torch.Tensor:
    """src: uint8 codes [n, rlen+slack]; returns codes [n, rlen] with substitutions, insertions
    (a random base that consumes no source base) and deletions (one source base skipped)."""
    n, w = src.shape
    dev = src.device
    u = torch.rand((n, rlen), generator=g, device=dev)
    is_ins = u < ins
    is_del = (u >= ins) & (u < ins + dele)
    step = 1 - is_ins.to(torch.int32) + is_del.to(torch.int32)
    consumed = torch.cumsum(step, dim=1) - step  # exclusive
    idx = (consumed + is_del.to(torch.int32)).clamp_(max=w - 1).long()
    out = torch.gather(src, 1, idx)
    rnd = torch.randint(0, 4, (n, rlen), generator=g, device=dev, dtype=torch.uint8)
    out = torch.where(is_ins, rnd, out)
    us = torch.rand((n, rlen), generator=g, device=dev)
    is_sub = (us < sub) & ~is_ins & (out < 4)
    shift = torch.randint(1, 4, (n, rlen), generator=g, device=dev, dtype=torch.uint8)
    out = torch.where(is_sub, (out + shift) % 4, out)
    if n_rate > 0:
        un = torch.rand((n, rlen), generator=g, device=dev)
        out = torch.where(un < n_rate, torch.full_like(out, 4), out)
    return out


def simulate_reads(donor: Genome, n: int, rlen: int, paired: bool, seed: int,
                   frag_mean: float = 500.0, frag_sd: float = 50.0, frag_min: int = 300, frag_max: int = 800,
                   sub: float = 0.005, ins: float = 0.001, dele: float = 0.001, n_rate: float = 0.0,
                   device: str = "cpu", chunk: int = 1 << 20, skip_head: int = 0,
                   skip_contigs: Sequence[int] = (0,)) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (bases, origin): ``bases`` uint8 ASCII [n_reads, rlen] (n_reads = 2n when paired,
    mates interleaved) and ``origin`` int64 [n_reads, 2] = (offset in the concatenated donor,
    strand) of each read's first base, for diagnostics only.

    ``skip_head`` keeps fragments out of the first bases of the first contig: within ~1.5x the
    insert-size estimate of the genome start the reference's mate rescue reads outside
    ``RefSequence`` (src/AlignmentRescue.cpp:87-93 computes ``PosDiff - EstDist`` < 0) and
    crashes, so golden vectors made with the real reference must avoid that region."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    cat = torch.cat([c.to(dev) for c in donor.codes])
    lens = torch.tensor([int(c.numel()) for c in donor.codes], dtype=torch.int64, device=dev)
    starts = torch.cumsum(lens, 0) - lens
    slack = 24
    ascii_lut = ASCII.to(dev)
    outs, origins = [], []
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        if paired:
            frag = (torch.randn(m, generator=g, device=dev) * frag_sd + frag_mean).round().long()
            frag = frag.clamp_(max(frag_min, rlen + slack), frag_max)
        else:
            frag = torch.full((m,), rlen + slack, dtype=torch.int64, device=dev)
        # contig chosen with probability ~ length, then a start inside it
        ci = torch.multinomial(lens.double(), m, replacement=True, generator=g)
        room = (lens[ci] - frag).clamp_(min=1)
        off = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * room.double()).long()
        if skip_head:
            first = torch.zeros_like(ci, dtype=torch.bool)
            for k in skip_contigs:  # donor contigs that are copies of the reference's first contig
                first |= ci == k
            off = torch.where(first, off.clamp(min=skip_head), off)
            off = torch.minimum(off, (lens[ci] - frag).clamp_(min=0))
        frag = torch.minimum(frag, lens[ci])
        fs = starts[ci] + off  # fragment start in the concatenation
        strand = torch.randint(0, 2, (m,), generator=g, device=dev)
        ar = torch.arange(rlen + slack, device=dev)
        # forward end of fragment and (reverse-complemented) far end
        head_idx = (fs[:, None] + ar[None, :]).clamp_(max=cat.numel() - 1)
        tail_idx = (fs[:, None] + frag[:, None] - 1 - ar[None, :]).clamp_(min=0)
        head = cat[head_idx]
        tail = cat[tail_idx]
        tail = torch.where(tail < 4, 3 - tail, tail)
        # strand 1: the fragment is read from the other side
        s1 = strand.bool()[:, None]
        first = torch.where(s1, tail, head)
        second = torch.where(s1, head, tail)
        r1 = _apply_errors(first, rlen, sub, ins, dele, n_rate, g)
        if paired:
            r2 = _apply_errors(second, rlen, sub, ins, dele, n_rate, g)
            both = torch.stack([r1, r2], dim=1).reshape(2 * m, rlen)
            o1 = torch.where(strand.bool(), fs + frag - 1, fs)
            o2 = torch.where(strand.bool(), fs, fs + frag - 1)
            org = torch.stack([torch.stack([o1, strand], 1), torch.stack([o2, 1 - strand], 1)], 1).reshape(2 * m, 2)
        else:
            both = r1
            org = torch.stack([torch.where(strand.bool(), fs + frag - 1, fs), strand], 1)
        outs.append(ascii_lut[both.long()])
        origins.append(org)
    return torch.cat(outs), torch.cat(origins)


def _records(bases: torch.Tensor, first: int, step: int, prefix: str, fastq: bool) -> np.ndarray:
    """Fixed-width FASTQ/FASTA records as one uint8 matrix (vectorised: tens of millions of reads)."""
    arr = bases[first::step].cpu().numpy()
    n, rlen = arr.shape
    digits = max(1, len(str(max(n - 1, 0))))
    head = (b"@" if fastq else b">") + prefix.encode() + b"_"
    width = len(head) + digits + 1 + rlen + 1 + (2 + rlen + 1 if fastq else 0)
    out = np.empty((n, width), dtype=np.uint8)
    c = 0
    out[:, c:c + len(head)] = np.frombuffer(head, dtype=np.uint8); c += len(head)
    idx = np.arange(n, dtype=np.int64)
    for d in range(digits):  # zero-padded decimal index
        out[:, c + digits - 1 - d] = 48 + (idx // (10 ** d)) % 10
    c += digits
    out[:, c] = 10; c += 1
    out[:, c:c + rlen] = arr; c += rlen
    out[:, c] = 10; c += 1
    if fastq:
        out[:, c] = 43; out[:, c + 1] = 10; c += 2
        out[:, c:c + rlen] = 73; c += rlen
        out[:, c] = 10
    return out


def write_fastq(path: str, bases: torch.Tensor, first: int, step: int, prefix: str = "sim", append: bool = False) -> None:
    """Rows first, first+step, ... of ``bases`` as FASTQ; read k of the file is named
    ``<prefix>_<k, zero padded>`` so that both mates of a pair share a name. Qualities are 'I'.
    ``append``: behind what the file holds (a further batch of reads under another prefix)."""
    rec = _records(bases, first, step, prefix, True)
    if append:
        with open(path, "ab") as fh:
            rec.tofile(fh)
    else:
        rec.tofile(path)


def write_fasta_reads(path: str, bases: torch.Tensor, first: int, step: int, prefix: str = "sim") -> None:
    """Same as write_fastq but FASTA records (no qualities).  Single-end parity tests use this:
    for reverse-strand single-end FASTQ records the reference prints a quality string whose first
    byte is uninitialised (src/SamReport.cpp:318-322)."""
    _records(bases, first, step, prefix, False).tofile(path)
