#!/usr/bin/env python3
"""Slot machinery under stress: thousands of small ragged batches through the three-slot device boundary
(mcx_stream_submit_packed -> mcx_stream_map32 -> mcx_stream_collect, a batch's copy in / packing under the kernels of the batch before it)
against the same batches through mcx_map_batch (host buffers, synchronous) on a second context with the same insert-size trajectory:
records and CIGAR words equal, batch by batch.  Kinds alternate — paired, single-end — and sizes change from batch to batch, so a slot is
reused by batches of other shapes and the pre-pack's guess (mated or not) is wrong half the time.

  MCX_PREPACK=1 python scripts/stress_slots.py --batches 2000        (the pre-pack on the copy stream)
  MCX_PREPACK=0 python scripts/stress_slots.py --batches 2000        (the default)

A batch that differs is written to --keep (reads, both record sets) and counted; exit code 1 when any did."""
import argparse
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapcaller_amd import api  # noqa: E402

FIELDS = ("pos", "mate_pos", "chr", "flag", "mapq", "tlen", "nm", "as", "xs", "n_cigar", "fwd", "has_mate")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--set", default="var")
    ap.add_argument("--max-batch", type=int, default=800)
    ap.add_argument("--keep", default=os.path.join(ROOT, "gpurun_out", "stress_fail"))
    ap.add_argument("--full-sa", type=int, default=2)
    a = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden", a.set)
    rd = lambda fn: [l for i, l in enumerate(gzip.open(os.path.join(gold, fn), "rb").read().split(b"\n")) if i % 4 == 1]
    r1, r2 = rd("r1.fq.gz"), rd("r2.fq.gz")
    rng = np.random.default_rng(a.seed)
    ix = api.Index(os.path.join(gold, "idx"), device=0, full_sa=a.full_sa)
    mp_s = api.Mapper(ix, alg="ksw2", max_batch_reads=a.max_batch)   # the stream side
    mp_h = api.Mapper(ix, alg="ksw2", max_batch_reads=a.max_batch)   # the synchronous side
    L = api.lib()
    code = np.zeros(256, dtype=np.uint32)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def draw(paired):
        n = int(rng.integers(1, a.max_batch // 200 + 1)) * 200 if paired else int(rng.integers(1, a.max_batch + 1))
        seqs = []
        for k in range(n):
            p = int(rng.integers(0, len(r1)))
            x = bytearray((r1 if (k & 1) == 0 else r2)[p] if paired else (r1 if rng.random() < 0.5 else r2)[p])
            if paired and (k & 1):
                x = bytearray(r2[seqs_pair[0]])
            if paired and not (k & 1):
                seqs_pair[0] = p
            if rng.random() < 0.3:
                x = x[: int(rng.integers(20, len(x) + 1))]
            if rng.random() < 0.04:
                x[int(rng.integers(0, len(x)))] = ord("N")
            if rng.random() < 0.02:
                j = int(rng.integers(0, len(x)))
                x[j] = ord(chr(x[j]).lower())
            seqs.append(bytes(x))
        return seqs

    seqs_pair = [0]
    bad = 0
    pend = []  # batches submitted and not yet compared: (kind, seqs, out buffers)
    outs = [(torch.zeros(a.max_batch * 32, dtype=torch.uint8).pin_memory(), torch.zeros(api.cigar_pool_words(a.max_batch), dtype=torch.int32).pin_memory()) for _ in range(3)]
    host = []  # pinned inputs of the batches in flight (kept alive until collected)

    def submit(seqs):
        row_words = (max(len(x) for x in seqs) + 15) // 16
        rows = np.zeros((len(seqs), row_words * 16), dtype=np.uint32)
        lens = np.array([len(x) for x in seqs], dtype=np.uint32)
        odd = []
        for r, x in enumerate(seqs):
            b = np.frombuffer(x, dtype=np.uint8)
            rows[r, : len(x)] = code[b]
            for pos in np.nonzero(~np.isin(b, acgt))[0]:
                odd.append((r << 32) | (int(pos) << 8) | int(b[pos]))
        words = (rows.reshape(len(seqs), row_words, 16).astype(np.uint64) << (30 - 2 * np.arange(16, dtype=np.uint64))).sum(-1).astype(np.uint32)
        tw = torch.from_numpy(words.astype(np.int64)).to(torch.int32).pin_memory()
        tl = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).pin_memory()
        to = torch.tensor(odd if odd else [0], dtype=torch.int64).pin_memory()
        host.append((tw, tl, to))
        assert L.mcx_stream_submit_packed(mp_s._h, tw.data_ptr(), row_words, tl.data_ptr(), len(seqs), to.data_ptr(), len(odd)) == 0, L.mcx_last_error()

    def check(k, paired, seqs, out):
        nonlocal bad
        n = len(seqs)
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(x) for x in seqs])
        w_aln, w_cig = mp_h.map_batch(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), off, paired)
        aln = api.aln32_unpack(np.frombuffer(out[0].numpy().tobytes(), dtype=api.ALN32_DTYPE)[:n])
        pool = out[1].numpy().view(np.uint32)
        ok = all(np.array_equal(aln[f], w_aln[f]) for f in FIELDS)
        if ok:
            for r in range(n):
                if not np.array_equal(pool[aln["cigar_off"][r]:aln["cigar_off"][r] + aln["n_cigar"][r]], w_cig[r]):
                    ok = False
                    break
        if not ok:
            bad += 1
            os.makedirs(a.keep, exist_ok=True)
            diff = {f: int((aln[f] != w_aln[f]).sum()) for f in FIELDS}
            with open(os.path.join(a.keep, f"seed{a.seed}_batch{k}.json"), "w") as fh:
                json.dump({"batch": k, "paired": paired, "reads": [s.decode("latin-1") for s in seqs], "fields_that_differ": diff,
                           "env": {e: v for e, v in os.environ.items() if e.startswith("MCX_")}}, fh)
            print(f"batch {k}: DIFFERS ({'paired' if paired else 'single'}, {n} reads) {diff}", flush=True)

    kinds, batches, total = [], [], 0
    while len(kinds) < a.batches:
        k = len(kinds)
        paired = (k % 3) != 1 if (k // 50) % 2 == 0 else bool(rng.integers(0, 2))  # stretches of P S P P S P ..., then stretches at random
        if paired and total % 200:  # (a paired batch starts on a 200-read chunk boundary of the run: a single-end batch of the missing reads in front)
            paired = False
            b = draw(False)[: 200 - total % 200]
            while len(b) < 200 - total % 200:
                b += draw(False)[: 200 - total % 200 - len(b)]
        else:
            b = draw(paired)
        kinds.append(paired)
        batches.append(b)
        total += len(b)
    # submit(i); map(i - 1); collect(i - 2) — the loop of INTEGRATION.md — with the comparison behind the collect
    for i in range(a.batches + 2):
        if i < a.batches:
            submit(batches[i])
        if 1 <= i <= a.batches:
            o = outs[(i - 1) % 3]
            assert L.mcx_stream_map32(mp_s._h, int(kinds[i - 1]), mp_s.avg, o[0].data_ptr(), o[1].data_ptr(), C.byref(mp_s.stats)) == 0, L.mcx_last_error()
        if i >= 2:
            assert L.mcx_stream_collect(mp_s._h, None, None) == 0, L.mcx_last_error()
            check(i - 2, kinds[i - 2], batches[i - 2], outs[(i - 2) % 3])
            host.pop(0)
    print(f"{a.batches - bad} of {a.batches} batches identical (MCX_PREPACK={os.environ.get('MCX_PREPACK', '')!r}, {sum(kinds)} paired, {a.batches - sum(kinds)} single-end)")
    mp_s.close(); mp_h.close(); ix.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
