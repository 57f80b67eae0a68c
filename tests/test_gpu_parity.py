"""Parity tests proper: the HIP path, called through the C ABI (libmcx.so), against the
reference's golden vectors and against the oracle on freshly seeded inputs.
Bar: bit-exact (integer / byte / index work)."""
import ctypes
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, SETS, VCF_CASES, VCF_RUNS, VcfOpts, maps_canon, sam_diff, vcf_alg, vcf_body

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from mapcaller_amd import api as a
    if not (os.path.exists(a.LIB_PATH) and os.path.exists(os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x"))):
        # a checkout without the built artefacts: build them here (hipcc is on the GPU box) — never a fallback
        subprocess.run(["make", "-C", os.path.join(ROOT, "mapcaller_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
    a.lib()  # raises if the HIP extension is missing: there is no fallback
    assert a.device_count() >= 1, "no GPU visible"
    return a


@pytest.fixture(scope="module")
def toy(api, golden):
    ix = api.Index(golden["toy"]["prefix"], device=0)
    yield ix
    ix.close()


def test_bwt_search_equals_reference_vectors(api, toy):
    q = json.load(open(os.path.join(GOLD, "func", "bwt_search.json")))
    mp = api.Mapper(toy, max_batch_reads=4096)
    seqs = [bytes("ACGTN".index(c) for c in r["seq"]) for r in q]
    ln, fr, loc = mp.bwt_search(seqs, [r["start"] for r in q])
    for i, r in enumerate(q):
        assert (int(ln[i]), int(fr[i])) == (r["len"], r["freq"]), i
        assert [int(x) for x in loc[i, :fr[i]]] == r["loc"], i
    mp.close()


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
def test_extend_equals_reference_vectors(api, toy, alg):
    cases = json.load(open(os.path.join(GOLD, "func", "dp.json")))
    mp = api.Mapper(toy, alg=alg, max_batch_reads=4096)
    ops, score = mp.extend(alg, [c["q"].encode() for c in cases], [c["t"].encode() for c in cases])
    for i, c in enumerate(cases):
        assert list(api.apply_ops(c["q"], c["t"], ops[i])) == c[alg], (i, c["q"], c["t"], ops[i])
        if alg == "ksw2":
            assert int(score[i]) == c["ksw2_score"], i
            assert ops[i][::-1] == c["ksw2_ops_rev"], i
    mp.close()


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", list(SETS))
@pytest.mark.parametrize("full_sa", [2, True, False])
def test_sam_equals_reference(api, golden, tmp_path, monkeypatch, name, alg, full_sa):
    """The reference's -t 1 SAM on every golden set, both algorithms.  full_sa (the product's default): every
    suffix-array entry in HBM, seeds leave k_seed as text positions; without it the sampled suffix array (k_sa); 2: the walk
    takes two bases per step over the pair records, which check themselves when they are made (a million random intervals
    extended both ways)."""
    g = golden[name]
    if full_sa == 2:
        monkeypatch.setenv("MCX_RANK2_CHECK", "1000000")
    ix = api.Index(g["prefix"], device=0, full_sa=full_sa)
    mp = api.Mapper(ix, alg=alg, max_batch_reads=1 << 14)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(g["r1"], g["r2"], out)
    assert st["reads"] > 0
    nd, ex = sam_diff(g["sam"][alg], out)
    assert nd == 0, ex
    mp.close(); ix.close()


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", list(SETS))
def test_sam_equals_reference_on_the_large_batch_paths(api, golden, tmp_path, monkeypatch, name, alg):
    """The golden sets again with what only a large batch switches on forced onto their small batches (MCX_ORDER_MIN=1): the straight-line
    pairs go from their seeds to their records in k_simple — small gaps aligned by the lane — and the rest is dealt to the per-pair kernels
    by weight; every DP list takes the one-problem-per-lane kernels (MCX_DP_LANE_ALWAYS=1), the long ones dealt by shape.  SAM identical
    to the reference's, and the straight-line path did take pairs."""
    monkeypatch.setenv("MCX_ORDER_MIN", "1")
    monkeypatch.setenv("MCX_DP_LANE_ALWAYS", "1")
    g = golden[name]
    ix = api.Index(g["prefix"], device=0, full_sa=True)  # (k_simple takes text positions: every suffix-array entry in HBM, the product's default)
    mp = api.Mapper(ix, alg=alg, max_batch_reads=1 << 14)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(g["r1"], g["r2"], out)
    nd, ex = sam_diff(g["sam"][alg], out)
    assert nd == 0, ex
    assert st["simple_pairs"] > (0 if name == "long" else 0.2 * st["reads"] / (2 if g["paired"] else 1)), st
    mp.close(); ix.close()


def test_overlapped_host_boundary_gives_the_same_records(api, golden, tmp_path):
    """mcx_stream_submit / map / collect (three batches in flight, copies on their own streams) against mcx_map_batch on the
    same batches: records and CIGAR words equal, read by read (the pool's offsets may differ)."""
    import torch
    g = golden["var"]
    reads1 = [l for i, l in enumerate(open(g["r1"], "rb").read().split(b"\n")) if i % 4 == 1]
    reads2 = [l for i, l in enumerate(open(g["r2"], "rb").read().split(b"\n")) if i % 4 == 1]
    n_pairs, per = 1000, 5
    batches = []
    for b in range(per):
        seqs = [x for p in range(b * n_pairs, (b + 1) * n_pairs) for x in (reads1[p], reads2[p])]
        off = np.zeros(len(seqs) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(x) for x in seqs])
        batches.append((np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), off))
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=2 * n_pairs)
    want = [mp.map_batch(bb, oo, True) for bb, oo in batches]
    mp.reset()
    L = api.lib()
    import ctypes as C
    pinned = []
    for bb, oo in batches:
        tb = torch.zeros(bb.size + 64, dtype=torch.uint8).pin_memory(); tb[:bb.size] = torch.from_numpy(bb)
        to = torch.from_numpy(oo.astype(np.int64)).to(torch.int32).pin_memory()  # same bits as uint32
        pinned.append((tb, to))
    outs = [(torch.zeros(2 * n_pairs * 64, dtype=torch.uint8).pin_memory(), torch.zeros(api.cigar_pool_words(2 * n_pairs), dtype=torch.int32).pin_memory()) for _ in range(per)]
    for i in range(per + 2):
        if i < per:
            assert L.mcx_stream_submit(mp._h, pinned[i][0].data_ptr(), pinned[i][1].data_ptr(), 2 * n_pairs) == 0, L.mcx_last_error()
        if 1 <= i <= per:
            assert L.mcx_stream_map(mp._h, 1, mp.avg, outs[i - 1][0].data_ptr(), outs[i - 1][1].data_ptr(), C.byref(mp.stats)) == 0, L.mcx_last_error()
        if i >= 2:
            assert L.mcx_stream_collect(mp._h, None, None) == 0, L.mcx_last_error()
    for b in range(per):
        aln = np.frombuffer(outs[b][0].numpy().tobytes(), dtype=api.ALN_DTYPE)
        pool = outs[b][1].numpy().view(np.uint32)
        w_aln, w_cig = want[b]
        for f in ("pos", "mate_pos", "chr", "flag", "mapq", "tlen", "nm", "as", "xs", "n_cigar", "fwd", "has_mate"):
            assert np.array_equal(aln[f], w_aln[f]), (b, f)
        for r in range(2 * n_pairs):
            assert np.array_equal(pool[aln["cigar_off"][r]:aln["cigar_off"][r] + aln["n_cigar"][r]], w_cig[r]), (b, r)
    mp.close(); ix.close()


def test_packed_host_boundary_gives_the_same_records(api, golden, tmp_path):
    """mcx_stream_submit_packed (2-bit rows + lengths + the list of bytes that are not ACGT, as the file front end's parser hands
    them over) against mcx_map_batch on the same reads as ASCII — ragged lengths, N, lower case and IUPAC letters sprinkled in:
    records and CIGAR words equal, read by read."""
    import ctypes as C
    import torch
    g = golden["var"]
    rng = np.random.default_rng(11)
    reads1 = [l for i, l in enumerate(open(g["r1"], "rb").read().split(b"\n")) if i % 4 == 1]
    reads2 = [l for i, l in enumerate(open(g["r2"], "rb").read().split(b"\n")) if i % 4 == 1]
    n_pairs, per = 1000, 4
    batches = []
    for b in range(per):
        seqs = []
        for pno in range(b * n_pairs, (b + 1) * n_pairs):
            for x in (reads1[pno], reads2[pno]):
                x = bytearray(x[: int(rng.integers(40, len(x) + 1))] if rng.random() < 0.2 else x)
                if rng.random() < 0.05:
                    x[int(rng.integers(0, len(x)))] = ord("N")
                if rng.random() < 0.03:
                    k = int(rng.integers(0, len(x)))
                    x[k] = ord(chr(x[k]).lower())
                if rng.random() < 0.01:
                    x[int(rng.integers(0, len(x)))] = ord("R")
                seqs.append(bytes(x))
        off = np.zeros(len(seqs) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(x) for x in seqs])
        batches.append((seqs, np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), off))
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=2 * n_pairs)
    want = [mp.map_batch(bb, oo, True) for _, bb, oo in batches]
    mp.reset()
    L = api.lib()
    code = np.zeros(256, dtype=np.uint32)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    packed = []
    for seqs, _, _ in batches:
        row_words = (max(len(x) for x in seqs) + 15) // 16
        rows = np.zeros((len(seqs), row_words * 16), dtype=np.uint32)
        lens = np.array([len(x) for x in seqs], dtype=np.uint32)
        odd = []
        for r, x in enumerate(seqs):
            a = np.frombuffer(x, dtype=np.uint8)
            rows[r, : len(x)] = code[a]
            for pos in np.nonzero(~np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8)))[0]:
                odd.append((r << 32) | (int(pos) << 8) | int(a[pos]))
        words = (rows.reshape(len(seqs), row_words, 16).astype(np.uint64) << (30 - 2 * np.arange(16, dtype=np.uint64))).sum(-1).astype(np.uint32)
        tw = torch.from_numpy(words.astype(np.int64)).to(torch.int32).pin_memory()
        tl = torch.from_numpy(lens.astype(np.int64)).to(torch.int32).pin_memory()
        to = torch.tensor(odd if odd else [0], dtype=torch.int64).pin_memory()
        packed.append((tw, row_words, tl, to, len(odd)))
    assert sum(p[4] for p in packed) > 50
    # (the records leave HBM as mcx_aln32 — 32 bytes each, packed on the device — and are unpacked here as mcx_aln_unpack does)
    outs = [(torch.zeros(2 * n_pairs * 32, dtype=torch.uint8).pin_memory(), torch.zeros(api.cigar_pool_words(2 * n_pairs), dtype=torch.int32).pin_memory()) for _ in range(per)]
    for i in range(per + 2):
        if i < per:
            tw, row_words, tl, to, n_odd = packed[i]
            assert L.mcx_stream_submit_packed(mp._h, tw.data_ptr(), row_words, tl.data_ptr(), 2 * n_pairs, to.data_ptr(), n_odd) == 0, L.mcx_last_error()
        if 1 <= i <= per:
            assert L.mcx_stream_map32(mp._h, 1, mp.avg, outs[i - 1][0].data_ptr(), outs[i - 1][1].data_ptr(), C.byref(mp.stats)) == 0, L.mcx_last_error()
        if i >= 2:
            assert L.mcx_stream_collect(mp._h, None, None) == 0, L.mcx_last_error()
    for b in range(per):
        aln = api.aln32_unpack(np.frombuffer(outs[b][0].numpy().tobytes(), dtype=api.ALN32_DTYPE))
        pool = outs[b][1].numpy().view(np.uint32)
        w_aln, w_cig = want[b]
        for f in ("pos", "mate_pos", "chr", "flag", "mapq", "tlen", "nm", "as", "xs", "n_cigar", "fwd", "has_mate"):
            assert np.array_equal(aln[f], w_aln[f]), (b, f)
        for r in range(2 * n_pairs):
            assert np.array_equal(pool[aln["cigar_off"][r]:aln["cigar_off"][r] + aln["n_cigar"][r]], w_cig[r]), (b, r)
    # a length the caller got wrong — longer than the read's row — is refused before anything is mapped (the lengths are checked on the
    # device, where the rows are turned back into bytes; nothing is written for such a read), and the context goes on with the next batch
    tw, row_words, tl, to, n_odd = packed[0]
    bad = tl.clone(); bad[7] = row_words * 16 + 1
    bad = bad.pin_memory()
    assert L.mcx_stream_submit_packed(mp._h, tw.data_ptr(), row_words, bad.data_ptr(), 2 * n_pairs, to.data_ptr(), n_odd) == 0, L.mcx_last_error()
    assert L.mcx_stream_map32(mp._h, 1, mp.avg, outs[0][0].data_ptr(), outs[0][1].data_ptr(), C.byref(mp.stats)) != 0
    assert b"longer than its row" in L.mcx_last_error()
    assert L.mcx_stream_submit_packed(mp._h, tw.data_ptr(), row_words, tl.data_ptr(), 2 * n_pairs, to.data_ptr(), n_odd) == 0, L.mcx_last_error()
    assert L.mcx_stream_map32(mp._h, 1, mp.avg, outs[0][0].data_ptr(), outs[0][1].data_ptr(), C.byref(mp.stats)) == 0, L.mcx_last_error()
    assert L.mcx_stream_collect(mp._h, None, None) == 0, L.mcx_last_error()
    # the same wrong length through the two halves (mcx_stream_next + mcx_map_batch_dev + mcx_stream_mapped32, what the file front end and the sharded
    # path call): the refusal arrives with the records, from mcx_stream_collect — round 5 returned 0 here with every read unmapped
    for fn in ("mcx_stream_next", "mcx_stream_mapped32"):
        getattr(L, fn).restype = C.c_int
    L.mcx_stream_next.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.mcx_stream_mapped32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    mp.reset()
    assert L.mcx_stream_submit_packed(mp._h, tw.data_ptr(), row_words, bad.data_ptr(), 2 * n_pairs, to.data_ptr(), n_odd) == 0, L.mcx_last_error()
    db, do, da, dc, nr = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
    assert L.mcx_stream_next(mp._h, C.byref(db), C.byref(do), C.byref(nr), C.byref(da), C.byref(dc)) == 0, L.mcx_last_error()
    mp.map_batch_dev(db.value, do.value, nr.value, True, da.value, dc.value)
    assert L.mcx_stream_mapped32(mp._h, outs[0][0].data_ptr(), outs[0][1].data_ptr()) == 0, L.mcx_last_error()
    assert L.mcx_stream_collect(mp._h, None, None) != 0
    assert b"longer than its row" in L.mcx_last_error() and b"empty reads" in L.mcx_last_error()
    # ... and what was said of that slot's buffers is said of no other batch: a device batch with an over-long read right after
    # mcx_stream_next (whose vouching the call above never took up) is still looked at
    mp.reset()
    assert L.mcx_stream_submit_packed(mp._h, tw.data_ptr(), row_words, tl.data_ptr(), 2 * n_pairs, to.data_ptr(), n_odd) == 0, L.mcx_last_error()
    assert L.mcx_stream_next(mp._h, C.byref(db), C.byref(do), C.byref(nr), C.byref(da), C.byref(dc)) == 0, L.mcx_last_error()
    long_off = torch.tensor([0, 400, 800], dtype=torch.int32, device="cuda")  # two reads of 400 bases in a context of 256
    long_bases = torch.full((800 + 64,), ord("A"), dtype=torch.uint8, device="cuda")
    with pytest.raises(api.McxError, match="longer than max_read_len"):
        mp.map_batch_dev(long_bases.data_ptr(), long_off.data_ptr(), 2, True, da.value, dc.value)
    mp.map_batch_dev(db.value, do.value, nr.value, True, da.value, dc.value)  # the slot's own batch, looked at like any other now
    assert L.mcx_stream_mapped32(mp._h, outs[0][0].data_ptr(), outs[0][1].data_ptr()) == 0, L.mcx_last_error()
    assert L.mcx_stream_collect(mp._h, None, None) == 0, L.mcx_last_error()
    aln = api.aln32_unpack(np.frombuffer(outs[0][0].numpy().tobytes(), dtype=api.ALN32_DTYPE))
    for f in ("pos", "flag", "mapq", "n_cigar"):
        assert np.array_equal(aln[f], want[0][0][f]), f
    mp.close(); ix.close()


@pytest.mark.parametrize("prepack", ["1", "0"])
def test_slots_under_stress_give_the_synchronous_paths_records(prepack):
    """scripts/stress_slots.py inside the suite: 500 small ragged batches — paired and single-end in turn, sizes changing from batch to batch, N and lower
    case sprinkled in — through the three-slot device boundary (copy in and unpacking of batch i + 1 under the kernels of batch i, copy out of batch i - 1
    beside them; MCX_PREPACK=1: packed for the kernels on the way in as well, with a guess about mates that is wrong half the time) against the same batches
    through mcx_map_batch on a second context: records and CIGAR words equal, batch by batch.  (the slot machinery that the
    pre-pack and the default path share; 2 x 2000 batches and 180 CLI fuzz rounds with the pre-pack on are profiles/round6/stress_slots.txt, fuzz_r6.txt.)"""
    env = dict(os.environ, MCX_PREPACK=prepack, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_slots.py"), "--batches", "500", "--seed", "23"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0 and "500 of 500 batches identical" in r.stdout, r.stdout[-2000:]


def test_small_batches_follow_the_avgdist_trajectory(api, golden, tmp_path):
    g = golden["mc"]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="nw", max_batch_reads=600)
    out = str(tmp_path / "gpu.sam")
    mp.map_files(g["r1"], g["r2"], out)
    nd, ex = sam_diff(g["sam"]["nw"], out)
    assert nd == 0, ex
    mp.close(); ix.close()


def _oracle_sam(prefix, f1, f2, alg, out):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmcx_oracle.so"))
    L.mcxo_index_load.restype = ctypes.c_void_p
    L.mcxo_index_load.argtypes = [ctypes.c_char_p]
    L.mcxo_map_files.restype = ctypes.c_int64
    L.mcxo_map_files.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
    ix = L.mcxo_index_load(prefix.encode())
    assert ix
    n = L.mcxo_map_files(ix, f1.encode(), (f2 or "").encode(), 0 if alg == "nw" else 1, out.encode(), 1, None)
    assert n > 0
    return n


def _against_reference(record_property, args, out_sam, tmp_path, **diff_kw):
    """The compiled reference (oracle/_ref/MapCaller -t 1) on the same input, where this box has it: its SAM against ours.
    Which checker judged the test is recorded (junit property `checker`, and a line on stdout): the oracle always has; the
    reference too unless the binary is absent or gives up on the input (it crashes on some degenerate reads)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    if not os.path.exists(ref_bin):
        which = "oracle only (no compiled reference on this box)"
    else:
        rs = str(tmp_path / "ref.sam")
        r = subprocess.run([ref_bin] + args + ["-sam", rs, "-no_vcf", "-t", "1", "-log", str(tmp_path / "job.log")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode != 0:
            which = f"oracle only (the reference exited with {r.returncode} on this input)"
        else:
            nd, ex = sam_diff(rs, out_sam, **diff_kw)
            assert nd == 0, ex
            which = "oracle and compiled reference"
    record_property("checker", which)
    print(f"[checker] {which}")
    return which


@pytest.mark.parametrize("alg,rlen,paired", [("ksw2", 150, True), ("nw", 250, True), ("ksw2", 100, False)])
def test_fresh_seeded_input_equals_oracle(api, tmp_path, record_property, alg, rlen, paired):
    """A 2 Mbp genome with repeats, 40 k reads: GPU SAM == oracle SAM (and == the compiled
    reference when oracle/_ref travelled to this box)."""
    from mapcaller_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    g = synth.random_genome([900000, 700000, 400000], seed=101 + rlen, n_repeats=60, repeat_len=800, tandem=30, n_runs=10)
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, g)
    prefix = str(tmp_path / "g")
    if os.path.exists(ref_bin):
        subprocess.run([ref_bin, "index", fa, prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        api.Index.build(fa, prefix, 0)
    donor = synth.mutate_genome(g, 7)
    n = 20000
    bases, _ = synth.simulate_reads(donor, n, rlen, paired, seed=5, skip_head=3000, sub=0.01, ins=0.002, dele=0.002, n_rate=0.0005)
    if paired:
        f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
        synth.write_fastq(f1, bases, 0, 2); synth.write_fastq(f2, bases, 1, 2)
    else:
        f1, f2 = str(tmp_path / "r1.fa"), None
        synth.write_fasta_reads(f1, bases, 0, 1)
    ix = api.Index(prefix, device=0)
    mp = api.Mapper(ix, alg=alg, max_read_len=max(256, rlen), max_batch_reads=1 << 15)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, f2, out)
    ora = str(tmp_path / "ora.sam")
    _oracle_sam(prefix, f1, f2, alg, ora)
    nd, ex = sam_diff(ora, out)
    assert nd == 0, ex
    which = _against_reference(record_property, ["-i", prefix, "-f", f1] + (["-f2", f2] if f2 else []) + ["-alg", alg], out, tmp_path)
    assert "exited" not in which, which  # (ordinary reads: the reference has no excuse here)
    assert st["mapped"] > 0.9 * st["reads"]
    mp.close(); ix.close()


@pytest.mark.parametrize("bucket_bases", ["", "1", "3"])
@pytest.mark.parametrize("name", ["toy", "mc", "long"])
def test_gpu_index_builder_writes_the_reference_files(api, golden, tmp_path, name, bucket_bases, monkeypatch):
    """mcx_index_build (GPU suffix sorting) against the five files `MapCaller index` wrote for the
    same FASTA: byte-identical (mc has runs of N: the lrand48 replacement path).  bucket_bases forces
    the bucketed sort that genome-scale texts (2^31 positions and more) take, on these small ones."""
    import gzip
    if bucket_bases:
        monkeypatch.setenv("MCX_BUILD_BUCKET_BASES", bucket_bases)
    fa = str(tmp_path / "genome.fa")
    open(fa, "wb").write(gzip.open(os.path.join(GOLD, name, "genome.fa.gz"), "rb").read())
    prefix = str(tmp_path / "built")
    api.Index.build(fa, prefix, 0)
    for ext in ("ann", "amb", "pac", "bwt", "sa"):
        a = open(f"{prefix}.{ext}", "rb").read()
        b = open(os.path.join(GOLD, name, f"idx.{ext}"), "rb").read()
        assert a == b, ext


def test_index_built_in_hbm_maps_like_the_loaded_one(api, golden, tmp_path):
    """Index.from_codes (genome already in HBM, full suffix array kept) -> same SAM, and its
    saved files equal the reference's."""
    import gzip
    import torch
    from mapcaller_amd import synth
    fa = str(tmp_path / "genome.fa")
    open(fa, "wb").write(gzip.open(os.path.join(GOLD, "long", "genome.fa.gz"), "rb").read())
    g = synth.read_fasta(fa)
    codes = torch.cat(g.codes).cuda()
    ix = api.Index.from_codes(codes.data_ptr(), [int(c.numel()) for c in g.codes], g.names, device=0, full_sa=True)
    prefix = str(tmp_path / "saved")
    ix.save(prefix)
    for ext in ("ann", "amb", "pac", "bwt", "sa"):
        assert open(f"{prefix}.{ext}", "rb").read() == open(os.path.join(GOLD, "long", f"idx.{ext}"), "rb").read(), ext
    gl = golden["long"]
    mp = api.Mapper(ix, alg="nw", max_batch_reads=1 << 12)
    out = str(tmp_path / "gpu.sam")
    mp.map_files(gl["r1"], gl["r2"], out)
    nd, ex = sam_diff(gl["sam"]["nw"], out)
    assert nd == 0, ex
    mp.close(); ix.close()


def test_reference_driver_linked_against_libmcx(golden, tmp_path):
    """The drop-in boundary itself: the reference's own main/ReadMapping/ReadAlignment/SamReport
    objects linked with integration/mapcaller_dropin.cpp (BWT_Search, nw_alignment,
    ksw2_alignment on top of libmcx.so) instead of bwt_search.o / nw_alignment.o /
    ksw2_alignment.o.  Built by `make -C oracle dropin` where the reference checkout exists."""
    exe = os.path.join(ROOT, "oracle", "_ref", "MapCaller_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/MapCaller_dropin was not built (no reference checkout at build time)")
    g = golden["toy"]
    for alg in ("ksw2", "nw"):
        out = str(tmp_path / f"dropin.{alg}.sam")
        cmd = [exe, "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-alg", alg, "-sam", out, "-no_vcf", "-t", "1", "-log", str(tmp_path / "job.log")]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        nd, ex = sam_diff(g["sam"][alg], out)
        assert nd == 0, ex


@pytest.mark.parametrize("straight_line", [False, True])
@pytest.mark.parametrize("name", list(SETS))
def test_alignment_profile_equals_reference(api, golden, tmp_path, monkeypatch, name, straight_line):
    """The -vcf bookkeeping on the GPU (k_prof_*): counter planes and sparse tallies against the
    reference's MappingRecordArr / InsertSeqMap / DeleteSeqMap / BreakPointMap / site lists.  straight_line: with what only a large
    batch switches on forced onto these small ones (MCX_ORDER_MIN=1) — the straight-line pairs' detail records then come from
    k_simple<.., DETAIL>, not from the finish stage."""
    import torch
    if straight_line:
        monkeypatch.setenv("MCX_ORDER_MIN", "1")
    g = golden[name]
    alg, prof, maps = g["prof"]
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    G = ix.genome_size
    mp = api.Mapper(ix, alg=alg, max_batch_reads=1000)  # several batches: the duplicate cap spans them
    planes = api.planes_alloc(G, "cuda")
    mp.profile_attach(planes.data_ptr())
    mp.map_files(g["r1"], g["r2"], None)
    mp.profile_finalize(planes.data_ptr())
    got = api.planes_view(planes, G).t().contiguous().cpu().numpy().astype(np.uint16)
    want = np.frombuffer(open(prof, "rb").read(), dtype=np.uint16).reshape(-1, 10)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (bad[:5], got[bad[:5, 0]], want[bad[:5, 0]])
    text = api.sparse_to_maps_text(mp.profile_sparse())
    assert maps_canon(text) == maps_canon(open(maps, encoding="latin-1").read())
    mp.close(); ix.close()


def test_profile_runs_equal_column_walk(api, golden, tmp_path, monkeypatch):
    """The -vcf bookkeeping writes what a read adds as runs (+1 / -1 differences, settled once) wherever the read's letters are
    plain; MCX_PROF_BY_COLUMN=1 makes it walk every fragment column by column instead (what it does anyway for reads with an odd
    letter).  Both ways on the `var` reads with letters lower-cased, N and IUPAC codes sprinkled in: the ten planes and the
    tally records must be the same — the runs, the settle and the odd-letter gate against the plain walk."""
    import gzip
    import torch
    g = golden["var"]
    rng = np.random.default_rng(11)

    def spoil(src, dst):
        op = gzip.open if src.endswith(".gz") else open
        lines = op(src, "rt").read().split("\n")
        for i in range(1, len(lines), 4):
            b = bytearray(lines[i].encode())
            if not b or rng.random() < 0.5:
                continue  # half of the reads stay plain
            for j in rng.integers(0, len(b), size=3):
                b[j] = ord(chr(b[j]).lower()) if rng.random() < 0.7 else ord("NRY"[int(rng.integers(0, 3))])
            lines[i] = b.decode()
        open(dst, "w").write("\n".join(lines))

    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    spoil(g["r1"], f1); spoil(g["r2"], f2)
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    res = []
    for by_column in (False, True):
        if by_column:
            monkeypatch.setenv("MCX_PROF_BY_COLUMN", "1")
        mp = api.Mapper(ix, alg="ksw2", max_batch_reads=4000)
        planes = api.planes_alloc(ix.genome_size, "cuda")
        mp.profile_attach(planes.data_ptr())
        mp.map_files(f1, f2, None)
        mp.profile_finalize(planes.data_ptr())
        sp = mp.profile_sparse_raw().copy()
        sp[:, 10:] *= (np.arange(54)[None, :] < sp[:, 9:10]).astype(np.uint8)
        res.append((api.planes_view(planes, ix.genome_size).cpu(), sorted(bytes(x) for x in sp)))
        mp.close()
    ix.close()
    assert int(res[0][0][0:4].sum()) > 100000
    for k in range(10):
        assert torch.equal(res[0][0][k], res[1][0][k]), api.PLANES[k]
    assert res[0][1] == res[1][1]


def test_profile_refuses_reads_after_the_settle(api, golden):
    """Once the differences have been turned into counts (profile_settle / profile_finalize) a further batch would add differences to
    counts: the context refuses it with a message until the profile is attached again — and maps without complaint after that."""
    import torch
    g = golden["toy"]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=4000)
    planes = api.planes_alloc(ix.genome_size, "cuda")
    mp.profile_attach(planes.data_ptr())
    mp.map_files(g["r1"], g["r2"], None)
    mp.profile_settle()
    mp.profile_settle()  # (idempotent)
    before = planes.clone()
    with pytest.raises(api.McxError, match="settled"):
        mp.map_files(g["r1"], g["r2"], None)
    assert torch.equal(planes, before)
    mp.profile_attach(planes.data_ptr())
    mp.reset()
    mp.map_files(g["r1"], g["r2"], None)
    mp.profile_finalize(planes.data_ptr())
    assert int(api.planes_view(planes, ix.genome_size)[0:4].sum()) > int(api.planes_view(before, ix.genome_size)[0:4].sum())
    mp.close(); ix.close()


@pytest.mark.parametrize("straight_line", [False, True])
@pytest.mark.parametrize("name,tag", VCF_CASES)
def test_vcf_equals_reference(api, golden, tmp_path, monkeypatch, name, tag, straight_line):
    """The whole -vcf surface on the GPU: mapping with the profile attached, then mcx_call_variants
    (k_vc_depth / k_vc_scan + sparse host logic) — the VCF of `MapCaller -vcf -t 1` line for line,
    for every switch the golden runs cover; straight_line: with k_simple (and its detail records) forced onto the small batches."""
    import torch
    if straight_line:
        monkeypatch.setenv("MCX_ORDER_MIN", "1")
    g = golden[name]
    o = VcfOpts(VCF_RUNS[tag][1]).struct
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    mp = api.Mapper(ix, alg=vcf_alg(name, tag), max_batch_reads=4000)
    planes = api.planes_alloc(ix.genome_size, "cuda")
    mp.profile_attach(planes.data_ptr(), max_dup=o.max_dup, max_clip=o.max_clip)
    st = mp.map_files(g["r1"], g["r2"], None)
    mp.profile_finalize(planes.data_ptr())
    out = str(tmp_path / "o.vcf")
    switches = {k: getattr(o, k) for k in ("ploidy", "min_allele_depth", "min_cnv", "min_gap", "fragment_size", "filter", "gvcf", "monomorphic", "somatic")}
    res = ix.call_variants(planes.data_ptr(), mp.profile_sparse(), st["pairs"], st["pair_dist_sum"], st["pair_len_sum"], out,
                           sample_id=o.sample_id.decode(), ref_name="ref", cmdline="test", **switches)
    got, want = vcf_body(out), vcf_body(g["vcf"][tag])
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad and len(got) == len(want), (len(got), len(want), bad[:3])
    assert res["n_records"] == sum(1 for l in want if l and not l.startswith("#"))
    mp.close(); ix.close()


def test_cli_sam_and_vcf(golden, tmp_path):
    """mapcaller-mi355x with the reference's command line (-i -f -f2 -alg -sam -vcf): both outputs
    equal the reference's from one run."""
    g = golden["var"]
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    cmd = [exe, "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-t", "4"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    nd, ex = sam_diff(g["sam"]["ksw2"], sam)
    assert nd == 0, ex
    assert vcf_body(vcf) == vcf_body(g["vcf"]["default"])


def test_run_is_fitted_to_the_hbm_that_is_left(api, golden, tmp_path):
    """mcx_ctx_create_fit (the -vcf leg left 5.4 GB of 309 and a larger genome met a bare hipMalloc failure): with most of the device
    declared out of reach (MCX_HBM_CAP_GB: the run may take what lies between a 128 K-read and a million-read context), a -vcf context for batches of a million reads over an index with pair records
    gives the pair records back first and then halves its batch until 4 GB stay free — said on stderr — and the CLI, which sizes its run the same
    way, still writes the golden SAM and VCF.  Without the cap nothing is degraded."""
    import torch
    g = golden["var"]
    ix = api.Index(g["prefix"], device=0, full_sa=2)
    plain = api.Mapper.fit_plan(ix, alg="ksw2", max_batch_reads=1 << 20, with_profile=True)
    assert plain["pair_records_trimmed"] == 0 and plain["batch_halvings"] == 0 and plain["max_batch_reads"] == 1 << 20 and plain["hbm_free_bytes"] > 4 << 30
    small = api.Mapper.fit_plan(ix, alg="ksw2", max_batch_reads=1 << 17, with_profile=True)
    assert 0 < small["hbm_taken_bytes"] < plain["hbm_taken_bytes"]
    # a cap between what batches of 128 K and of a million reads take (a context has a fixed part — DP scratch, the large tier's records — of well over 10 GB)
    cap_gb = ((small["hbm_taken_bytes"] + plain["hbm_taken_bytes"]) / 2 + (4 << 30)) / (1 << 30)
    os.environ["MCX_HBM_CAP_GB"] = "%.2f" % cap_gb
    try:
        plan = api.Mapper.fit_plan(ix, alg="ksw2", max_batch_reads=1 << 20, with_profile=True)
    finally:
        os.environ.pop("MCX_HBM_CAP_GB", None)
    assert plan["pair_records_trimmed"] == 1 and plan["batch_halvings"] >= 1, plan
    assert (1 << 17) <= plan["max_batch_reads"] < (1 << 20) and plan["max_batch_reads"] % 200 == 0 and plan["hbm_free_bytes"] >= 4 << 30, plan
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=plan["max_batch_reads"])  # (the trimmed index maps as before: one base per step)
    out = str(tmp_path / "gpu.sam")
    mp.map_files(g["r1"], g["r2"], out)
    nd, ex = sam_diff(g["sam"]["ksw2"], out)
    assert nd == 0, ex
    mp.close(); ix.close()
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    env = dict(os.environ, MCX_HBM_CAP_GB="%.2f" % cap_gb)  # (the CLI's default batch is 2 M reads)
    r = subprocess.run([exe, "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-t", "4", "-two_base"], stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-1000:]
    assert "[mcx fit]" in r.stderr and "pair records back" in r.stderr and "reads instead" in r.stderr, r.stderr[-1500:]
    nd, ex = sam_diff(g["sam"]["ksw2"], sam)
    assert nd == 0, ex
    assert vcf_body(vcf) == vcf_body(g["vcf"]["default"])
    # a device with no room at all: refused in words, not in the middle of a batch
    env = dict(os.environ, MCX_HBM_CAP_GB="%.2f" % (small["hbm_taken_bytes"] / 2 / (1 << 30)))  # (less than a context's fixed part)
    r = subprocess.run([exe, "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-sam", sam, "-vcf", vcf], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "no room for this run" in r.stderr, r.stderr[-800:]


def test_cli_builds_its_index_from_the_reference_fasta(golden, tmp_path):
    """-r ref.fa (main.cpp:217, :344-349): the CLI builds the index for the run on the GPU, maps against it and removes it again.
    The toy set's genome is the reference's test/ref.fa: SAM and VCF must be the golden run's, and nothing may stay behind in the
    temporary directory — neither after the run nor after a run that fails once the index exists (an unreadable read file)."""
    g = golden["toy"]
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    fa = str(tmp_path / "genome.fa")
    open(fa, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "toy", "genome.fa.gz"), "rb").read())
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, TMPDIR=str(scratch))
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    cmd = [exe, "-r", fa, "-f", g["r1"], "-f2", g["r2"], "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-t", "4"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, env=env)
    nd, ex = sam_diff(g["sam"]["ksw2"], sam)
    assert nd == 0, ex
    assert vcf_body(vcf) == vcf_body(g["vcf"]["default"])
    assert os.listdir(scratch) == [], os.listdir(scratch)
    # an error after the build: the index files are removed on that way out too
    r = subprocess.run([exe, "-r", fa, "-f", str(tmp_path / "missing.fq"), "-sam", str(tmp_path / "x.sam"), "-no_vcf"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode != 0, r.stderr[-500:]
    assert os.listdir(scratch) == [], os.listdir(scratch)
    # and a FASTA that cannot be indexed at all
    r = subprocess.run([exe, "-r", str(tmp_path / "missing.fa"), "-f", g["r1"], "-no_vcf"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode != 0
    assert os.listdir(scratch) == [], os.listdir(scratch)


def test_input_side_cases(api, io_golden, tmp_path):
    """mcx_map_files_ex against the reference's readers (GetData.cpp): interleaved -p input with an odd
    tail chunk and decorated headers, multi-line FASTA, .gz files handed over as they are."""
    g = io_golden
    ix = api.Index(g["prefix"], device=0)
    for alg, args, kw, ref in (("ksw2", (g["il.fq"], None), {"interleaved": True}, "ref.il.sam"),
                               ("nw", (g["ml.fa"], None), {}, "ref.ml.sam"),
                               ("ksw2", (g["gz1"], g["gz2"]), {"threads": 3}, "ref.gz.sam")):
        mp = api.Mapper(ix, alg=alg, max_batch_reads=1000)
        out = str(tmp_path / (ref + ".out"))
        mp.map_files(args[0], args[1], out, **kw)
        nd, ex = sam_diff(g[ref], out, mask_se_reverse_qual=True)
        assert nd == 0, (ref, ex)
        mp.close()
    # ordinary .gz files once more: through zlib's one thread (MCX_GZ_SERIAL=1, the reader of rounds 1-5) and — the default above — through the
    # parallel reader (mcx_pgz.h); and a larger pair of ordinary .gz files (the `var` reads at gzip -6, stretches of 64 KB so that every round
    # searches for block starts and fills windows in) against the golden SAM
    os.environ["MCX_GZ_SERIAL"] = "1"
    try:
        mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1000)
        out = str(tmp_path / "serial.gz.out")
        mp.map_files(g["gz1"], g["gz2"], out, threads=3)
        nd, ex = sam_diff(g["ref.gz.sam"], out, mask_se_reverse_qual=True)
        assert nd == 0, ("MCX_GZ_SERIAL", ex)
        mp.close()
    finally:
        os.environ.pop("MCX_GZ_SERIAL", None)
    ix.close()


def test_plain_gz_pairs_through_the_parallel_reader(api, golden, tmp_path):
    """The `var` pairs as ordinary gzip files (levels 1, 6, 9): the SAM of the plain files."""
    import gzip
    g = golden["var"]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1 << 13)
    for level in (1, 6, 9):
        f = []
        for k in ("r1", "r2"):
            f.append(str(tmp_path / f"{k}.l{level}.fq.gz"))
            with gzip.open(f[-1], "wb", compresslevel=level) as fh:
                fh.write(open(g[k], "rb").read())
        out = str(tmp_path / f"l{level}.sam")
        mp.reset()
        st = mp.map_files(f[0], f[1], out)
        nd, ex = sam_diff(g["sam"]["ksw2"], out)
        assert nd == 0, (level, ex)
        assert st["reads"] == 2 * open(g["r1"], "rb").read().count(b"\n") // 4
    mp.close(); ix.close()


def test_cli_two_libraries(io_golden, tmp_path):
    """-f a1 b1 -f2 a2 b2: one SAM stream, one header, the insert-size estimate carried from the first
    library into the second (the reference's globals)."""
    g = io_golden
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam = str(tmp_path / "lib.sam")
    cmd = [exe, "-i", g["prefix"], "-f", g["a1"], g["b1"], "-f2", g["a2"], g["b2"], "-alg", "ksw2", "-sam", sam, "-no_vcf"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    nd, ex = sam_diff(g["ref.lib.sam"], sam)
    assert nd == 0, ex


def test_eight_shards_with_full_batches_equal_single_stream(tmp_path):
    """BASELINE config 4's shape on one GPU: the native CLI as one stream and as EIGHT shards (-devices 0,0,0,0,0,0,0,0) with batches
    of 512 K reads — every slot of the per-round exchanges (insert-size sums, duplicate-check keys, SAM places) filled with a full
    batch, 10x coverage so that the duplicate cap decides — over 2 M pairs on a 60 Mbp human-like genome (eight contexts and eight
    sets of planes share this box's HBM): the 1.5 GB SAM and the VCF must be the same bytes (scripts/shard_scale.py).
    (Placed ahead of the tests that keep the full-size index in HBM: the eight contexts need the room.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "shard_scale.py"), "--genome-mbp", "60", "--contigs", "4", "--shards", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, env=dict(os.environ, PYTHONPATH=ROOT), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert o["runs"]["single"]["rc"] == 0 and o["runs"]["shards"]["rc"] == 0, o
    assert o["sam_identical"] and o["vcf_identical"], o
    assert o["runs"]["shards"]["sam_bytes"] > 1_000_000_000 and o["vcf_records"] > 0


@pytest.fixture(scope="module")
def bench_genome(api, tmp_path_factory):
    """bench.py's workload at full size: the GRCh38-sized (3.1 Gbp, 6.2 G text positions) synthetic genome with its human-like repeat
    landscape, indexed on the GPU (the bucketed index builder, the 15-mer jump table, the full suffix array in HBM) and saved for
    the CPU checker.  Shared by the tests that need it: building it takes most of a minute."""
    import argparse
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(genome_mbp=3100.0, contigs=24, repeats=2000, genome="human")  # bench.py's default
    codes, lens, _ = bench.make_genome(args, dev, seed=1234)
    os.environ["MCX_RANK2_CHECK"] = "4000000"  # (the pair records of bench.py's default index check themselves as they are made)
    try:
        ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=2)
    finally:
        os.environ.pop("MCX_RANK2_CHECK", None)
    prefix = str(tmp_path_factory.mktemp("big") / "big")
    ix.save(prefix)
    yield {"codes": codes, "lens": lens, "index": ix, "prefix": prefix, "bench": bench, "dev": dev}
    ix.close()


@pytest.fixture(scope="module")
def full_size_reference(bench_genome, tmp_path_factory):
    """ONE run of the reference at full size for the two tests that compare with it (configs 3 and 4 share the index and the reads; the reference spends
    45 s loading the 3.1 Gbp index and maps 20 k reads a second at -t 1): 300 k pairs x 150 bp of the bench workload through its own main() with -sam and
    -vcf, the profile and maps dumped between Mapping() and VariantCalling() (mcref_tool R).  None where the compiled reference did not travel."""
    from mapcaller_amd import synth
    ref_tool = os.path.join(ROOT, "oracle", "_ref", "mcref_tool")
    if not os.path.exists(ref_tool):
        yield None
        return
    g = bench_genome
    d = tmp_path_factory.mktemp("fullref")
    n_pairs = 300_000
    reads = g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 150, seed=1000, device=g["dev"]).reshape(2 * n_pairs, 150).cpu()
    f1, f2 = str(d / "r1.fq"), str(d / "r2.fq")
    synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
    ref_sam, ref_vcf, dump = str(d / "ref.sam"), str(d / "ref.vcf"), str(d / "ref")
    r = subprocess.run([ref_tool], input=f"R ksw2 {dump} {g['prefix']} {ref_sam} {ref_vcf} {f1} {f2}\n", text=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=2000)
    assert r.stdout.strip().split("\n")[-1].strip() == "ok", r.stdout[-300:]
    yield {"f1": f1, "f2": f2, "n_pairs": n_pairs, "sam": ref_sam, "vcf": ref_vcf, "dump": dump}


def _checker_sam(prefix, f1, f2, alg, out, tmp_path):
    """The compiled reference at -t 1 when it travelled to this box, else the oracle restatement."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    if os.path.exists(ref_bin):
        cmd = [ref_bin, "-i", prefix, "-f", f1, "-f2", f2, "-alg", alg, "-sam", out, "-no_vcf", "-t", "1", "-log", str(tmp_path / "job.log")]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
    else:
        _oracle_sam(prefix, f1, f2, alg, out)


def test_full_size_genome_prefix_equals_reference(api, bench_genome, full_size_reference, tmp_path, record_property):
    """BASELINE config 3 at full size: the first 300 k pairs of a bench batch (150 bp, -alg ksw2) as ONE batch of 600 k reads through the
    product's file path — everything only a large batch switches on: the straight-line pairs through k_simple, the others dealt to the
    lanes by weight, the large tier beside tier 0, mate rescue beside the build, the late pairs' pass, the batch's tail queued behind its
    kernels — against the compiled reference's `-t 1` SAM of the same reads (full_size_reference: one run of the reference for this test and
    config 4's; a whole bench batch of 4 M pairs against it is scripts/full_batch_parity.py, a quarter of an hour of its clock).  Reads from
    repeats bring hundreds of seed hits, mate rescue and the large-capacity tier with them (asserted).  The insert-size trajectory of a
    prefix is the trajectory of the run, so the SAM must be identical.  Without the compiled reference: 60 k pairs against the oracle."""
    from mapcaller_amd import synth
    g = bench_genome
    if full_size_reference:
        f1, f2, n_pairs, chk = full_size_reference["f1"], full_size_reference["f2"], full_size_reference["n_pairs"], full_size_reference["sam"]
    else:
        n_pairs = 60_000
        reads = g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 150, seed=1000, device=g["dev"]).reshape(2 * n_pairs, 150).cpu()
        f1, f2, chk = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), str(tmp_path / "chk.sam")
        synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
        _oracle_sam(g["prefix"], f1, f2, "ksw2", chk)
    mp = api.Mapper(g["index"], alg="ksw2", max_batch_reads=2 * n_pairs)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, f2, out)  # (a batch of the file path is the context's max_batch_reads: the whole input here)
    mp.close()
    nd, ex = sam_diff(chk, out)
    assert nd == 0, ex
    assert st["reads"] == 2 * n_pairs and st["mapped"] > 0.95 * st["reads"]
    assert st["tier1_pairs"] > 0, st  # pairs over the tier-0 capacities did go through the large tier
    assert n_pairs < 100_000 or st["simple_pairs"] > 0.3 * n_pairs, st  # the straight-line path took its share
    record_property("pairs", n_pairs)
    record_property("checker", "compiled reference" if full_size_reference else "oracle")


def _nonzero_plane_records(planes, G):
    """(positions, [n, 10] uint16) of the genome positions where any of the ten finalized planes is non-zero, in pieces (a genome-sized
    temporary does not fit beside the planes)."""
    import torch
    from mapcaller_amd import api as a
    pos, val = [], []
    for lo in range(0, G, 1 << 27):
        hi = min(G, lo + (1 << 27))
        v = a.planes_view(planes, G, lo, hi)                       # [10, hi - lo] int32
        idx = (v != 0).any(0).nonzero().reshape(-1)
        pos.append((idx + lo).cpu().numpy())
        val.append(v[:, idx].t().contiguous().cpu().numpy().astype(np.uint16))
        del v, idx
    return np.concatenate(pos), np.concatenate(val)


def test_config4_vcf_slice_at_full_size_equals_reference(api, bench_genome, full_size_reference, tmp_path, record_property):
    """BASELINE config 4's per-GPU slice at full size: 300 k pairs x 150 bp of the bench workload against the 3.1 Gbp genome with the
    -vcf bookkeeping on — profile attached, several batches (the duplicate cap spans them), pair records resident — through the product's
    file path, then mcx_call_variants; against the compiled reference's own `-t 1 -sam -vcf` run on the same index files and its
    MappingRecordArr / maps after Mapping() (mcref_tool R: its main() once, the positions with a non-zero counter dumped before VariantCalling()).  SAM, the ten planes at every non-zero
    position (positions above 2^31 among them: asserted), the insert / delete / break-point maps and site lists, and the VCF body must
    be identical.  (main.cpp:372 new MappingRecord_t[GenomeSize], AlignmentProfile.cpp:41-271, VariantCalling.cpp:696-740.)"""
    import torch
    if not full_size_reference:
        pytest.skip("the compiled reference did not travel to this box (the oracle's dense profile does not fit a 3.1 Gbp genome)")
    g = bench_genome
    ix = g["index"]
    G = ix.genome_size
    f1, f2, ref_sam, ref_vcf, dump = (full_size_reference[k] for k in ("f1", "f2", "sam", "vcf", "dump"))
    want = np.fromfile(dump + ".prof.nz", dtype=np.dtype([("pos", "<i8"), ("v", "<u2", (10,))]))
    # the product: one context, batches of 128 K reads
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1 << 17)
    planes = api.planes_alloc(G, "cuda:0")
    mp.profile_attach(planes.data_ptr())
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, f2, out)
    mp.profile_finalize(planes.data_ptr())
    nd, ex = sam_diff(ref_sam, out)
    assert nd == 0, ex
    got_pos, got_val = _nonzero_plane_records(planes, G)
    assert got_pos.size == want.size and np.array_equal(got_pos, want["pos"]), (got_pos.size, want.size)
    bad = np.argwhere(got_val != want["v"])
    assert bad.size == 0, (bad[:5], got_pos[bad[:5, 0]], got_val[bad[:5, 0]], want["v"][bad[:5, 0]])
    assert int((got_pos > (1 << 31)).sum()) > 1_000_000, "no counters above 2^31"
    sparse = mp.profile_sparse()
    assert maps_canon(api.sparse_to_maps_text(sparse)) == maps_canon(open(dump + ".maps", encoding="latin-1").read())
    vcf = str(tmp_path / "gpu.vcf")
    res = ix.call_variants(planes.data_ptr(), sparse, st["pairs"], st["pair_dist_sum"], st["pair_len_sum"], vcf, ref_name="ref", cmdline="test")
    got, wantv = vcf_body(vcf), vcf_body(ref_vcf)
    badv = [(a, b) for a, b in zip(got, wantv) if a != b]
    assert not badv and len(got) == len(wantv), (len(got), len(wantv), badv[:3])
    record_property("checker", "compiled reference")
    print(f"[config 4 slice] {st['reads']} reads, {got_pos.size} positions with counters, {len(sparse)} tally records, {res['n_records']} VCF records: all equal to the reference's")
    mp.close()
    del planes
    torch.cuda.empty_cache()


def test_config2_ecoli_sized_single_end_equals_reference(api, tmp_path):
    """BASELINE config 2 at its own size: an E. coli-sized genome (4.6 Mbp, one contig, bench.py's generator), 300 k single-end
    reads x 100 bp (FASTA: the reference prints a stray quality byte for reverse-strand single-end FASTQ), -alg ksw2, index built on
    the GPU and saved for the CPU checker — the compiled reference at -t 1 when it is on this box.  SAM line for line."""
    import argparse
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from mapcaller_amd import synth
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(genome_mbp=4.6, contigs=1, repeats=20, genome="uniform")
    codes, lens, _ = bench.make_genome(args, dev, seed=1234)
    ix = api.Index.from_codes(codes.data_ptr(), lens, device=0, full_sa=2)
    prefix = str(tmp_path / "ecoli")
    ix.save(prefix)
    n = 300000
    reads = bench.make_reads(codes, lens, n, 100, seed=2002, device=dev, paired=False).reshape(n, 100).cpu()
    f1 = str(tmp_path / "r.fa")
    synth.write_fasta_reads(f1, reads, 0, 1)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1 << 17)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, None, out)
    mp.close(); ix.close()
    chk = str(tmp_path / "chk.sam")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "MapCaller")
    if os.path.exists(ref_bin):
        subprocess.run([ref_bin, "-i", prefix, "-f", f1, "-alg", "ksw2", "-sam", chk, "-no_vcf", "-t", "1", "-log", str(tmp_path / "job.log")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
    else:
        _oracle_sam(prefix, f1, None, "ksw2", chk)
    nd, ex = sam_diff(chk, out)
    assert nd == 0, ex
    assert st["reads"] == n and st["mapped"] > 0.97 * n


def test_large_batch_machinery_does_not_change_the_records(api, bench_genome, monkeypatch):
    """One large batch of the bench workload (120 k pairs on the full-size genome) mapped three times: with everything that only a large
    batch switches on — the straight-line pairs through k_simple, the others dealt to the lanes by weight, the large tier beside tier 0 on
    its own streams, mate rescue beside the build of the pairs that do not await it, the late list's pass on the third set of resources
    taking its seed hits from tier 0 —, the same with every DP list on the one-problem-per-lane kernels, and with all of it off (one pass
    after the other on one stream, the wavefront DP kernels: what the small-batch tests compare with the reference).  Records and CIGAR
    words must be equal read by read."""
    g = bench_genome
    n_pairs = 120000
    reads = g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 150, seed=77, device=g["dev"]).reshape(2 * n_pairs, 150).cpu().numpy()
    bases = np.ascontiguousarray(reads).reshape(-1)
    off = (np.arange(2 * n_pairs + 1, dtype=np.uint64) * 150).astype(np.uint32)

    def run():
        mp = api.Mapper(g["index"], alg="ksw2", max_batch_reads=2 * n_pairs)
        aln, cig = mp.map_batch(bases, off, True)
        st = dict(tier1=mp.stats.tier1_pairs, simple=mp.stats.simple_pairs)
        mp.close()
        return aln, cig, st

    a_aln, a_cig, a_st = run()
    assert a_st["simple"] > 0.3 * n_pairs, a_st  # the straight-line path took its share
    monkeypatch.setenv("MCX_DP_LANE_ALWAYS", "1")  # every DP list one problem per lane (at this size the long lists take the wavefront kernels)
    c_aln, c_cig, c_st = run()
    monkeypatch.delenv("MCX_DP_LANE_ALWAYS")
    # the large tier's build: every pair handed to one lane inside k_build_wave (what it does by itself for a pair whose fragment bounds do
    # not fit the pool — stage_build's dense placement and overflow test)
    monkeypatch.setenv("MCX_BUILD_WAVE_LIMIT", "0")
    d_aln, d_cig, d_st = run()
    monkeypatch.delenv("MCX_BUILD_WAVE_LIMIT")
    # everything off: no k_simple, no order, the wavefront DP kernels, the rescue in line, the late pairs searched again, the large tier's build a
    # lane per pair, the tiers one after the other, the seeding walk one base per step
    for k in ("MCX_NO_WORK_ORDER", "MCX_NO_LATE_OVERLAP", "MCX_NO_TIER_OVERLAP", "MCX_NO_SIMPLE", "MCX_DP_BY_WAVE", "MCX_RESCUE_IN_LINE", "MCX_LATE_RESEED", "MCX_BUILD_BY_LANE",
              "MCX_SEED_ONE_BASE"):
        monkeypatch.setenv(k, "1")
    b_aln, b_cig, b_st = run()
    assert a_st["tier1"] > 0 and a_st["tier1"] == b_st["tier1"] == c_st["tier1"] == d_st["tier1"], (a_st, b_st, c_st, d_st)
    assert b_st["simple"] == 0, b_st
    for x_aln, x_cig in ((b_aln, b_cig), (c_aln, c_cig), (d_aln, d_cig)):
        for f in ("pos", "mate_pos", "chr", "flag", "mapq", "tlen", "nm", "as", "xs", "n_cigar", "fwd", "has_mate"):
            assert np.array_equal(a_aln[f], x_aln[f]), f
        for r in range(2 * n_pairs):
            assert np.array_equal(a_cig[r], x_cig[r]), r


def test_bench_workload_keeps_its_shape(api, bench_genome):
    """Guard rails for the numbers bench.py reports, on its own workload at a quarter of its batch (1 M pairs x 150 bp on the
    full-size genome, reads resident in HBM): the shares of work that decide the step — pairs sent to the large-capacity tier,
    pairs replayed for the insert-size estimate, index blocks and DP problems per read — stay where they were measured, and a
    steady-state step stays under a bound twice its measured time (a change that doubles the step fails here, not in a
    hand-run bench)."""
    import time
    import torch
    g = bench_genome
    n_pairs = 1_000_000
    n = 2 * n_pairs
    dev = g["dev"]
    batches = [g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 150, seed=4242 + k, device=dev).reshape(-1).contiguous() for k in range(3)]
    off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * 150).to(torch.uint32)
    d_aln = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_cig = torch.empty(api.cigar_pool_words(n), dtype=torch.int32, device=dev)
    mp = api.Mapper(g["index"], alg="ksw2", max_batch_reads=n)
    mp.map_batch_dev(batches[0].data_ptr(), off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())  # first use: allocations, the estimate's first pairs
    before = mp.stats.as_dict()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[1:]:
        mp.map_batch_dev(b.data_ptr(), off.data_ptr(), n, True, d_aln.data_ptr(), d_cig.data_ptr())
    torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / 2
    after = mp.stats.as_dict()
    d = {k: after[k] - before[k] for k in after}
    mp.close()
    reads = d["reads"]
    print(f"[bench shape] {ms:.2f} ms per {n} reads; tier 1 {d['tier1_pairs']}, replayed {d['replayed_pairs']}, index blocks/read {d['fm_blocks'] / reads:.2f}, "
          f"DP problems/read {d['dp_jobs'] / reads:.3f}, mapped {d['mapped'] / reads:.4f}")
    assert reads == 2 * n
    assert d["mapped"] > 0.97 * reads, d
    assert 0 < d["tier1_pairs"] < 0.02 * (reads / 2), d           # measured 0.7-0.8 %
    assert d["replayed_pairs"] < 0.02 * (reads / 2), d           # steady state: the estimate barely moves
    assert d["halved_selections"] == 0, d
    assert 6 < d["fm_blocks"] / reads < 30, d                     # measured 19-20 index records per read one base per step
    assert d["dp_jobs"] / reads < 1.0, d                          # measured 0.3-0.4 DP problems per read
    assert ms < 16.0, f"{ms:.2f} ms per 2 M reads (measured 8.5 ms)"


def test_config5_indel_heavy_long_pairs_equal_reference(api, bench_genome, tmp_path, monkeypatch):
    """BASELINE config 5 read literally: 250 bp pairs with 5 % indels per base (2.5 % insertions + 2.5 % deletions), -alg nw, against the
    full-size index, 100 k pairs in ONE batch — several gapped fragments per read.  The DP job lists are held to 100 k entries here
    (MCX_JOB_CAP; at bench.py's batch size they run over by themselves), so the pass runs over and the selection is mapped in
    halves (asserted).  The reference's gates leave few of these reads mapped; mostly-unmapped output is the expected answer,
    line for line."""
    from mapcaller_amd import synth
    g = bench_genome
    n_pairs = 100_000
    reads = g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 250, seed=77, device=g["dev"], sub=0.005, ins=0.025, dele=0.025).reshape(2 * n_pairs, 250).cpu()
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
    monkeypatch.setenv("MCX_JOB_CAP", "100000")
    mp = api.Mapper(g["index"], alg="nw", max_read_len=256, max_batch_reads=2 * n_pairs)
    monkeypatch.delenv("MCX_JOB_CAP")
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, f2, out)
    mp.close()
    chk = str(tmp_path / "chk.sam")
    _checker_sam(g["prefix"], f1, f2, "nw", chk, tmp_path)
    nd, ex = sam_diff(chk, out)
    assert nd == 0, ex
    assert st["dp_jobs"] > 2 * st["reads"], st       # config 5's character: several DP problems per read
    assert st["halved_selections"] > 0, st           # the job lists did run over and the batch was mapped in halves
    assert 0 < st["mapped"] < 0.5 * st["reads"], st


def test_large_tier_records_and_dp_scratch_grow_with_the_batches(api, bench_genome, tmp_path, monkeypatch, capfd):
    """A batch that sends the large tier more pairs than its records hold makes them grow (tier1_grow: config 5 at bench.py's size has 185 k such pairs
    and started with room for 99 k — the second pass ran alone after tier 0), and a batch whose 65-256-column DP list is long makes that list's scratch
    grow for the batches after it (dp_scratch_grow).  160 k pairs of config 5's kind in two batches, 0.5 GB of records to start with (2 k pairs) and the
    list's threshold lowered: the SAM of the run that grows equals, byte for byte, the one of a run that keeps what it started with; each growth is
    reported once; a device made to look full (MCX_HBM_CAP_GB) keeps what it has and still maps the same."""
    from mapcaller_amd import synth
    g = bench_genome
    n_pairs = 160_000
    reads = g["bench"].make_reads(g["codes"], g["lens"], n_pairs, 250, seed=78, device=g["dev"], sub=0.005, ins=0.025, dele=0.025).reshape(2 * n_pairs, 250).cpu()
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, reads, 0, 2); synth.write_fastq(f2, reads, 1, 2)
    monkeypatch.setenv("MCX_ALLOC_LOG", "1")
    monkeypatch.setenv("MCX_DP_GROW_MIN", "1000")
    outs, logs, stats = [], [], []
    for mode in ("grow", "fixed", "full"):
        monkeypatch.delenv("MCX_TIER1_GB", raising=False); monkeypatch.delenv("MCX_HBM_CAP_GB", raising=False); monkeypatch.delenv("MCX_NO_TIER1_GROW", raising=False)
        monkeypatch.setenv("MCX_TIER1_START_GB", "0.5")
        if mode == "fixed":
            monkeypatch.delenv("MCX_TIER1_START_GB"); monkeypatch.setenv("MCX_TIER1_GB", "1"); monkeypatch.setenv("MCX_NO_TIER1_GROW", "1")
        if mode == "full":
            monkeypatch.setenv("MCX_HBM_CAP_GB", "1")
        capfd.readouterr()
        mp = api.Mapper(g["index"], alg="nw", max_read_len=256, max_batch_reads=n_pairs)  # two batches of 80 k pairs
        out = str(tmp_path / f"{mode}.sam")
        stats.append(mp.map_files(f1, f2, out))
        mp.close()
        logs.append(capfd.readouterr().err)
        outs.append(open(out, "rb").read())
    assert stats[0]["tier1_pairs"] > 2 * 2100, stats[0]  # more heavy pairs a batch than the 0.5 GB holds
    assert logs[0].count("the large tier's records grow") == 1, logs[0][-2000:]
    assert logs[0].count("DP list's scratch grows") == 1, logs[0][-2000:]
    assert "grow" not in logs[1] and "grow" not in logs[2], (logs[1][-1000:], logs[2][-1000:])
    assert outs[0] == outs[1] == outs[2]
    assert len(outs[0]) > 10_000_000


def test_fuzz_rounds_equal_oracle():
    """scripts/fuzz_parity.py inside the suite, bounded: 24 rounds with fixed seeds (the long runs are scripts/fuzz_r6.sh's; a round that differs leaves its files in gpurun_out/fuzz_fail) — random genomes (contigs, repeats, tandem and N runs),
    donors, read lengths 36-300, single / paired, FASTQ / FASTA, error rates up to 5 % substitutions and 1 % indels, both algorithms, the
    variant-calling switches — the CLI's SAM and VCF against the oracle's, line by line."""
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--rounds", "24", "--seed", "2027"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "24 of 24 rounds identical" in r.stdout


def test_fuzz_rounds_on_three_shards_equal_oracle():
    """The same generator with the reads dealt to three shards (mapcaller-mi355x -devices 0,0,0, batches of 400 reads): 16 rounds — the
    oracle is a single stream, so every round checks the shards' exchange (insert-size trajectory, duplicate cap, discordant-pair events)."""
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--rounds", "16", "--seed", "909", "--cli-args", "-devices 0,0,0 -batch 400"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "16 of 16 rounds identical" in r.stdout


def test_fuzz_rounds_on_the_large_batch_paths_equal_oracle(monkeypatch):
    """The generator once more, mapping alone (`-no_vcf`: no alignment profile is kept, so the straight-line path is open to the pairs; the
    index with its pair records) with what only a large batch switches on forced onto the small ones — k_simple with its DP problems
    collected, solved and replayed, the order lists, every DP list on the lane kernels (two problems per lane): 20 rounds, the SAM against the oracle's."""
    monkeypatch.setenv("MCX_ORDER_MIN", "1")
    monkeypatch.setenv("MCX_DP_LANE_ALWAYS", "1")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--rounds", "20", "--seed", "5150", "--no-vcf"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "20 of 20 rounds identical" in r.stdout


def test_fuzz_rounds_on_the_large_batch_paths_with_the_profile_equal_oracle(monkeypatch):
    """The same with -vcf: the alignment profile is kept, the straight-line pairs' detail records come from k_simple (mcx_simple.h
    SimpleDetail) and the others' from the finish stage: 10 rounds, SAM and VCF against the oracle's."""
    monkeypatch.setenv("MCX_ORDER_MIN", "1")
    monkeypatch.setenv("MCX_DP_LANE_ALWAYS", "1")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--rounds", "10", "--seed", "31337"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1400)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "10 of 10 rounds identical" in r.stdout


def test_overlong_read_is_refused(api, golden):
    """One read longer than the context's max_read_len in a batch handed over through the C ABI: refused before any kernel touches the
    per-read slots (the file front end names the read; here the batch fails as a whole)."""
    g = golden["toy"]
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    mp = api.Mapper(ix, alg="ksw2", max_read_len=128, max_batch_reads=64)
    seqs = [b"ACGT" * 25, b"ACGT" * 25, b"ACGT" * 40, b"ACGT" * 25]  # 100, 100, 160 (> 128), 100 bases
    off = np.zeros(5, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in seqs])
    with pytest.raises(api.McxError, match="longer than max_read_len"):
        mp.map_batch(np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(), off, True)
    ok = [b"ACGT" * 25] * 4  # the context is still usable
    off[1:] = np.cumsum([len(x) for x in ok])
    aln, _ = mp.map_batch(np.frombuffer(b"".join(ok), dtype=np.uint8).copy(), off, True)
    assert len(aln) == 4
    mp.close(); ix.close()


def test_ragged_reads_equal_oracle(api, tmp_path, record_property):
    """Read lengths from 12 to 300 in one batch (mates trimmed independently), some below the 16-base
    seed minimum, Ns sprinkled in: GPU SAM == oracle SAM (== the reference when it is here and survives)."""
    from mapcaller_amd import synth
    rng = np.random.default_rng(77)
    g = synth.random_genome([600000, 300000], seed=303, n_repeats=30, repeat_len=700, tandem=10, n_runs=6)
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, g)
    prefix = str(tmp_path / "g")
    api.Index.build(fa, prefix, 0)
    n = 6000
    bases, _ = synth.simulate_reads(synth.mutate_genome(g, 9), n, 300, True, seed=8, skip_head=3000, frag_mean=700, frag_sd=80, frag_min=350,
                                    frag_max=1000, sub=0.01, ins=0.002, dele=0.002, n_rate=0.003)
    arr = bases.cpu().numpy()
    lens = rng.integers(12, 301, size=arr.shape[0])
    lens[rng.random(arr.shape[0]) < 0.5] = 150
    for k, path in ((0, "r1.fq"), (1, "r2.fq")):
        with open(tmp_path / path, "wb") as fh:
            for i in range(k, arr.shape[0], 2):
                s = arr[i, : lens[i]].tobytes()
                fh.write(b"@rag_%05d\n%s\n+\n%s\n" % (i // 2, s, b"F" * len(s)))
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    ix = api.Index(prefix, device=0)
    mp = api.Mapper(ix, alg="ksw2", max_read_len=320, max_batch_reads=4000)
    out = str(tmp_path / "gpu.sam")
    st = mp.map_files(f1, f2, out)
    ora = str(tmp_path / "ora.sam")
    _oracle_sam(prefix, f1, f2, "ksw2", ora)
    nd, ex = sam_diff(ora, out)
    assert nd == 0, ex
    _against_reference(record_property, ["-i", prefix, "-f", f1, "-f2", f2, "-alg", "ksw2"], out, tmp_path)
    assert st["mapped"] > 0.8 * st["reads"]
    mp.close(); ix.close()


def test_long_cigars_equal_oracle(api, tmp_path, record_property):
    """Reads with a one-base deletion or insertion every 17 bases (exact 16-mers between them): the
    alignments need 33+ CIGAR operations, more than a read's average share of the batch's CIGAR pool
    (MCX_CIGAR_STRIDE words per read + MCX_CIGAR_SLACK) — next to ordinary indel-heavy 300 bp reads
    (BASELINE config 5's regime), and then by themselves in batches of ONE pair, where nothing but the
    pool's slack holds their operations.  GPU SAM == oracle SAM (== the reference when it is here)."""
    import re
    import torch
    from mapcaller_amd import synth
    g = synth.random_genome([700000, 300000], seed=404, n_repeats=10, repeat_len=500)
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, g)
    prefix = str(tmp_path / "g")
    api.Index.build(fa, prefix, 0)
    bases, _ = synth.simulate_reads(g, 3000, 300, True, seed=12, skip_head=3000, frag_mean=700, frag_sd=60, frag_min=400, frag_max=1000,
                                    sub=0.001, ins=0.02, dele=0.02)
    chrom = g.codes[0].numpy()
    rng = np.random.default_rng(5)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    extra = []
    for k in range(300):  # mate 1: 17 blocks of 17 bases with one base dropped (or added) between blocks; mate 2: exact, reverse strand
        p0 = int(rng.integers(5000, 690000))
        blocks, p = [], p0
        for j in range(17):
            blocks.append(chrom[p:p + 17])
            p += 17
            if k % 2 == 0:
                p += 1                                      # deletion in the read
            elif j < 16:
                blocks.append(np.array([(chrom[p] + 1) % 4], dtype=np.uint8))  # inserted base that differs from the next one
        r1 = np.concatenate(blocks)[:300]
        r1 = np.concatenate([r1, chrom[p:p + 300 - len(r1)]]) if len(r1) < 300 else r1
        m = chrom[p0 + 500:p0 + 800]
        r2 = (3 - m)[::-1]
        extra += [lut[r1], lut[r2]]
    bases = torch.cat([bases, torch.from_numpy(np.stack(extra))])
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, bases, 0, 2); synth.write_fastq(f2, bases, 1, 2)
    for alg in ("nw", "ksw2"):
        ix = api.Index(prefix, device=0)
        mp = api.Mapper(ix, alg=alg, max_read_len=320, max_batch_reads=10000)
        out = str(tmp_path / f"gpu.{alg}.sam")
        mp.map_files(f1, f2, out)
        ora = str(tmp_path / f"ora.{alg}.sam")
        _oracle_sam(prefix, f1, f2, alg, ora)
        longest = max((len(re.findall(r"\d+[MIDS]", l.split("\t")[5])) for l in open(ora) if not l.startswith("@")), default=0)
        assert longest > 32, longest  # the case this test exists for
        nd, ex = sam_diff(ora, out)
        assert nd == 0, ex
        mp.close(); ix.close()
    _against_reference(record_property, ["-i", prefix, "-f", f1, "-f2", f2, "-alg", "ksw2"], str(tmp_path / "gpu.ksw2.sam"), tmp_path)
    # the long-CIGAR pairs alone, ONE pair per batch (mcx_map_batch on a context sized for two reads): the operations of a read
    # with 33+ of them fit only because the pool has slack beyond its per-read share; the CIGARs are the file run's
    sam_lines = [l.split("\t") for l in open(str(tmp_path / "gpu.ksw2.sam")) if not l.startswith("@")]
    ix = api.Index(prefix, device=0)
    mp = api.Mapper(ix, alg="ksw2", max_read_len=320, max_batch_reads=2)
    n_long = 0
    for k in range(0, 40, 2):
        pair = np.stack(extra[k:k + 2])
        mp.reset()  # (every call a run of its own: a batch starts on a chunk boundary)
        aln, cig = mp.map_batch(np.ascontiguousarray(pair).reshape(-1), np.array([0, 300, 600], dtype=np.uint32), True)
        text = "".join(f"{int(w) >> 4}{'MID?S'[int(w) & 15]}" for w in cig[0])
        want = sam_lines[2 * (3000 + k // 2)][5]
        if int(aln["chr"][0]) >= 0 and want != "*":
            assert text == want, (k, text, want)
            n_long += len(cig[0]) > 32
    assert n_long >= 5, n_long
    mp.close(); ix.close()


def _write_bgzf(path, data, block=0xff00, level=6):
    """bgzip's container: independent gzip members of at most 64 KB, each carrying its own size in a 'BC' extra field."""
    import struct
    import zlib
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            chunk = data[i:i + block]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_bgzf_input_equals_plain(api, golden, tmp_path):
    """FASTQ in bgzip's container (independent members, inflated side by side by the reader's threads) gives the SAM the plain
    and the ordinary .gz files give — the `var` pairs, members of 64 KB and of 3 KB (records and lines cut by member borders
    everywhere); a file whose tail is not BGZF ends there like a damaged gzip stream does."""
    import gzip
    g = golden["var"]
    raw = [gzip.open(g[k], "rb").read() if g[k].endswith(".gz") else open(g[k], "rb").read() for k in ("r1", "r2")]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1 << 15)
    for block in (0xff00, 3000):
        f1, f2 = str(tmp_path / f"b{block}_1.fq.gz"), str(tmp_path / f"b{block}_2.fq.gz")
        _write_bgzf(f1, raw[0], block); _write_bgzf(f2, raw[1], block)
        out = str(tmp_path / f"b{block}.sam")
        mp.reset()
        st = mp.map_files(f1, f2, out)
        nd, ex = sam_diff(g["sam"]["ksw2"], out)
        assert nd == 0, (block, ex)
        assert st["reads"] == 2 * raw[0].count(b"\n") // 4
    # the first half of the members, then bytes that are no member: the reads of the first half come out, nothing fails
    half = str(tmp_path / "half_1.fq.gz")
    whole = open(str(tmp_path / "b3000_1.fq.gz"), "rb").read()
    cut = whole.find(b"\x1f\x8b\x08\x04", len(whole) // 2)
    open(half, "wb").write(whole[:cut] + b"not a member at all, forty bytes of it..")
    mp.reset()
    st = mp.map_files(half, None, str(tmp_path / "half.sam"))
    assert 0 < st["reads"] < raw[0].count(b"\n") // 4
    mp.close(); ix.close()


def test_one_context_takes_paired_then_single_end_batches(api, golden, tmp_path):
    """A context keeps one pair record per PAIR of its largest paired batch and grows to one per read only when a single-end batch
    needs that (tier 0's records are the largest thing a context owns).  One context sized for 8 000 reads maps the `var` pairs,
    then mate 1 of them as single-end reads in batches of 8 000 (more reads than it has pair records: the growth), then the pairs
    again: every SAM equals its checker's."""
    g = golden["var"]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=8000)
    ora_se = str(tmp_path / "ora.se.sam")
    _oracle_sam(g["prefix"], g["r1"], None, "ksw2", ora_se)
    for k, (f1, f2, want, kw) in enumerate(((g["r1"], g["r2"], g["sam"]["ksw2"], {}), (g["r1"], None, ora_se, {"mask_se_reverse_qual": True}),
                                            (g["r1"], g["r2"], g["sam"]["ksw2"], {}))):
        out = str(tmp_path / f"o{k}.sam")
        mp.reset()
        mp.map_files(f1, f2, out)
        nd, ex = sam_diff(want, out, **kw)
        assert nd == 0, (k, ex)
    mp.close(); ix.close()


def test_file_path_errors_are_loud(api, golden, tmp_path):
    """No fallbacks: a missing read file, a mate file that holds fewer reads, a read longer than the
    context was sized for — each is an error with a message, not a shorter SAM."""
    g = golden["toy"]
    ix = api.Index(g["prefix"], device=0)
    mp = api.Mapper(ix, alg="ksw2", max_read_len=256, max_batch_reads=1000)
    with pytest.raises(api.McxError, match="cannot open"):
        mp.map_files(str(tmp_path / "nope.fq"), None, None)
    lines = open(g["r2"], "rb").read().split(b"\n")
    short = tmp_path / "short.fq"
    short.write_bytes(b"\n".join(lines[:4 * 700]) + b"\n")
    with pytest.raises(api.McxError, match="fewer reads"):
        mp.map_files(g["r1"], str(short), str(tmp_path / "x.sam"))
    mp.close()
    mp = api.Mapper(ix, alg="ksw2", max_read_len=100, max_batch_reads=1000)
    with pytest.raises(api.McxError, match="max_read_len"):
        mp.map_files(g["r1"], g["r2"], None)
    mp.close(); ix.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world,batch", [(2, 2000), (3, 1000), (2, 400)])
def test_sharded_run_equals_single_stream(golden, tmp_path, world, batch):
    """mapcaller_amd.run under torchrun with several ranks (all on GPU 0, gloo so that they can share it):
    batches dealt to the ranks in turn, one insert-size trajectory and one duplicate-cap order kept
    across the shards by the per-round exchange, SAM parts merged back into input order, planes reduced
    onto rank 0, sparse tallies and discordant-pair events gathered, variants called on rank 0.
    SAM *and* VCF equal the reference's single-stream (-t 1) run on the `var` set (SNVs, indels, an
    inversion, a moved segment; ~30x: the duplicate cap bites, the first 1000 pairs move the estimate)."""
    g = golden["var"]
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "mapcaller_amd.run", "-backend", "gloo", "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"],
           "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-batch", str(batch)]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
    nd, ex = sam_diff(g["sam"]["ksw2"], sam)
    assert nd == 0, ex
    assert vcf_body(vcf) == vcf_body(g["vcf"]["default"])
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


@pytest.mark.parametrize("devices,batch,name,alg", [("0,0", "2000", "var", "ksw2"), ("0,0,0", "600", "var", "ksw2"), ("0,0", "400", "se", "ksw2"),
                                                   ("0,0,0", "400", "mc", "nw")])
def test_native_cli_several_shards_equals_single_stream(golden, tmp_path, devices, batch, name, alg):
    """mapcaller-mi355x -devices a,b,..: one host thread and one context per shard inside one process
    (the shards share GPU 0 here; the profile reduce then takes its in-process path instead of RCCL),
    the exchange between host threads, the SAM parts merged by the C++ side.  Same bar: single-stream SAM and VCF,
    on a paired set, a single-end set and a multi-contig set."""
    g = golden[name]
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    cmd = [exe, "-i", g["prefix"], "-f", g["r1"]] + (["-f2", g["r2"]] if g["r2"] else []) + ["-alg", alg, "-sam", sam, "-vcf", vcf, "-t", "2",
                                                                                            "-devices", devices, "-batch", batch]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
    nd, ex = sam_diff(g["sam"][alg], sam, mask_se_reverse_qual=not g["r2"])
    assert nd == 0, ex
    if vcf_alg(name, "default") == alg:
        assert vcf_body(vcf) == vcf_body(g["vcf"]["default"])
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


def test_native_cli_shards_two_libraries(io_golden, tmp_path):
    """Two libraries over two shards: one SAM stream, the insert-size state carried from the first library into
    the second on every shard (the reference's globals)."""
    g = io_golden
    exe = os.path.join(ROOT, "mapcaller_amd", "mapcaller-mi355x")
    sam = str(tmp_path / "lib.sam")
    cmd = [exe, "-i", g["prefix"], "-f", g["a1"], g["b1"], "-f2", g["a2"], g["b2"], "-alg", "ksw2", "-sam", sam, "-no_vcf", "-devices", "0,0", "-batch", "400"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    nd, ex = sam_diff(g["ref.lib.sam"], sam)
    assert nd == 0, ex


def test_profile_reduce_over_rccl_one_rank(api, golden, monkeypatch):
    """libmcx_comm.so on the GPU box: a one-rank communicator through RCCL's own entry points (ncclGetUniqueId,
    ncclCommInitRank, ncclReduce) — what a single-GPU box can exercise of the N-GPU reduce."""
    import ctypes
    import torch
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", "lo")  # (a one-rank bootstrap has no business probing the box's other interfaces)
    monkeypatch.setenv("NCCL_IB_DISABLE", "1")
    L = ctypes.CDLL(api.COMM_LIB_PATH)
    ident = (ctypes.c_uint8 * 128)()
    assert L.mcx_comm_unique_id(ident) == 0, api.lib().mcx_last_error()
    comm = ctypes.c_void_p()
    assert L.mcx_comm_init_rank(ident, 0, 1, 0, ctypes.byref(comm)) == 0, api.lib().mcx_last_error()
    G = 100_000
    rows = (torch.arange(10 * G, dtype=torch.int64, device="cuda").reshape(10, G) % 70001).to(torch.int32)
    rows[0:4] %= 9000    # A C G T: a count stays below 15 000 (mcx_planes.h); some above the 12-bit field
    rows[6:10] %= 60000  # strand counters that can travel two to a word on one rank
    planes = api.planes_from_rows(rows)
    secs = ctypes.c_double()
    L.mcx_profile_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    assert L.mcx_profile_reduce(comm, planes.data_ptr(), G, 0, ctypes.byref(secs)) == 0, api.lib().mcx_last_error()
    torch.cuda.synchronize()
    # the 16-bit planes travel as the words they lie in (A C G T clamped to 4095 first): what comes back equals the input once the
    # field widths are applied, multi_hit and readCount untouched
    got = api.planes_view(planes, G)
    assert torch.equal(got[0:4], rows[0:4].clamp(max=4095))
    assert torch.equal(got[4], rows[4]) and torch.equal(got[5], rows[5] & 0xFFFF)
    assert torch.equal(got[6:10], rows[6:10])
    # the same through the scattered form of the reduce (what three ranks and more take: slices exchanged all to all, summed by their owners,
    # gathered onto the root — with one rank: its own slice through the same kernels and groups)
    monkeypatch.setenv("MCX_REDUCE_SCATTER", "1")
    planes = api.planes_from_rows(rows)
    assert L.mcx_profile_reduce(comm, planes.data_ptr(), G, 0, ctypes.byref(secs)) == 0, api.lib().mcx_last_error()
    torch.cuda.synchronize()
    assert torch.equal(api.planes_view(planes, G), got)
    monkeypatch.delenv("MCX_REDUCE_SCATTER")
    # strand counters that could carry into the neighbouring half on several ranks (here: times one rank still fits, so force the wide path
    # with a value at the top of the range on a communicator of one) — and the wide path itself: one counter per word, narrowed again
    rows[6] = 0xFFFF
    monkeypatch.setenv("MCX_REDUCE_WIDE", "1")
    planes = api.planes_from_rows(rows)
    assert L.mcx_profile_reduce(comm, planes.data_ptr(), G, 0, ctypes.byref(secs)) == 0, api.lib().mcx_last_error()
    torch.cuda.synchronize()
    got = api.planes_view(planes, G)
    assert torch.equal(got[6:10], rows[6:10]) and torch.equal(got[0:4], rows[0:4].clamp(max=4095)) and torch.equal(got[4], rows[4])
    L.mcx_comm_free.argtypes = [ctypes.c_void_p]
    L.mcx_comm_free(comm)


def test_run_module_single_gpu(golden, tmp_path):
    """python -m mapcaller_amd.run without torchrun: one GPU, the same outputs as the reference."""
    g = golden["var"]
    sam, vcf = str(tmp_path / "o.sam"), str(tmp_path / "o.vcf")
    cmd = [sys.executable, "-m", "mapcaller_amd.run", "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-alg", "ksw2", "-sam", sam, "-vcf", vcf, "-gvcf"]
    subprocess.run(cmd, check=True, env=dict(os.environ, PYTHONPATH=ROOT), cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    nd, ex = sam_diff(g["sam"]["ksw2"], sam)
    assert nd == 0, ex
    assert vcf_body(vcf) == vcf_body(g["vcf"]["gvcf"])


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_bench_launches_its_ranks(tmp_path, n_ranks):
    """`python bench.py --gpus N` the way the driver runs it on a node with several GPUs — the launcher starts one rank per GPU before
    any GPU call, the ranks walk one insert-size trajectory per step — with the ranks sharing this box's one GPU over gloo and a
    small genome: one JSON line, N GPUs' worth of reads, the exchange inside the timed region.  N = 8 is the node the driver's
    scaling run uses: its first 8-rank run is not the first one ever."""
    env = dict(os.environ, PYTHONPATH=ROOT, MCX_BENCH_SHARE_GPU="1", MCX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    pairs = 100000 if n_ranks == 2 else 40000
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--genome", "uniform", "--genome-mbp", "20", "--contigs", "4",
           "--repeats", "50", "--batch-pairs", str(pairs), "--cpu-pairs", "0", "--vcf-reduce", "1", "--pcie-steps", "0", "--second-genome", "0", "--other-configs", "0",
           "--file-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    if r.returncode != 0:  # (the launcher's summary fills the tail of stderr: the failing rank's own words are further up — kept whole where the box's files come back from)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", f"bench_ranks_{n_ranks}.stderr.txt"), "w").write(r.stderr)
    assert r.returncode == 0, [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Traceback" in l][:12] + [r.stderr[-1500:]]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    o = json.loads(lines[0])
    assert o["n_gpus"] == n_ranks and o["steps"] == 2 and o["scaling"] == "weak"
    assert o["value"] > 0 and abs(o["value"] - n_ranks * 2 * 2 * pairs / (o["ms_per_step"] * 2 / 1000)) < 0.01 * o["value"]
    assert o["simple_pairs"] > 0  # (the straight-line path runs under the per-rank trajectory steps too)
    assert o["per_read"]["mapped_frac"] > 0.9
    assert "exchanges" in o["config"]["multi_gpu"]
    assert o["config"]["multi_gpu_host_ms_per_step"] is not None
    # the -vcf leg over the two ranks: differences settled, the planes summed onto rank 0 as they lie (22 bytes per position here: the
    # readCount plane of independent runs is summed too), variants called there
    v = o["vcf_reduce"]
    assert "error" not in v, v
    # (two ranks: one reduce per piece, 22 bytes per position; eight: the pieces scattered, the root sends seven eighths of them)
    assert abs(v["reduce_gb"] - 22 * 20e6 / 1e9 * (1 if n_ranks == 2 else (n_ranks - 1) / n_ranks)) < 0.02 and v["call_variants_records"] > 0 and v["covered_positions"] > 1_000_000 * (1 if n_ranks == 2 else 3)
    # the line is the compact record the driver parses (round 5's 33.6 KB line came back unparsed); everything else is in the detail record it names
    assert len(lines[0]) < 6000 and "per_kernel" not in o["roofline"]
    detail = json.load(open(os.path.join(ROOT, o["detail"])))
    assert detail["value"] == o["value"] and "per_kernel" in detail["roofline"] and detail["vcf_reduce"]["call_variants"]["records"] == v["call_variants_records"]


def test_degenerate_reads_equal_oracle(api, golden, tmp_path, record_property):
    """Reads the path has little to say about — shorter than a seed, all N, homopolymers and short
    tandem repeats (more than 50 occurrences: BWT_Search reports none), a read that is the genome's
    first / last bases — next to ordinary ones: GPU SAM == oracle SAM (== the reference if it survives)."""
    import gzip
    g = golden["mc"]
    genome = b"".join(l for l in gzip.open(os.path.join(GOLD, "mc", "genome.fa.gz"), "rb").read().split(b"\n") if not l.startswith(b">"))
    rng = np.random.default_rng(3)
    reads = []
    for n in (1, 5, 15, 16, 17, 31, 40):
        p = int(rng.integers(5000, 100000))
        reads.append(genome[p:p + n])
    reads += [b"N" * 150, b"A" * 150, b"ACACACACAC" * 15, b"N" * 20 + genome[40000:40110] + b"N" * 20, genome[7000:7150].lower()]
    reads += [genome[3000:3150], genome[len(genome) - 150:], genome[len(genome) - 151:len(genome) - 1]]
    for k in range(40):
        p = int(rng.integers(5000, 150000))
        reads.append(genome[p:p + 150])
    def rc(s):
        return s[::-1].translate(bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN"))
    with open(tmp_path / "r1.fq", "wb") as f1, open(tmp_path / "r2.fq", "wb") as f2:
        for i, s in enumerate(reads):
            s = s.replace(b"n", b"N")
            mate = rc(genome[60000 + 300 * i: 60000 + 300 * i + 150])
            f1.write(b"@d%03d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))
            f2.write(b"@d%03d\n%s\n+\n%s\n" % (i, mate, b"I" * len(mate)))
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    ix = api.Index(g["prefix"], device=0)
    for paired in (True, False):
        for alg in ("ksw2", "nw"):
            mp = api.Mapper(ix, alg=alg, max_batch_reads=400)
            out = str(tmp_path / f"gpu.{alg}.{paired}.sam")
            mp.map_files(f1, f2 if paired else None, out)
            ora = str(tmp_path / f"ora.{alg}.{paired}.sam")
            _oracle_sam(g["prefix"], f1, f2 if paired else None, alg, ora)
            nd, ex = sam_diff(ora, out, mask_se_reverse_qual=True)
            assert nd == 0, (alg, paired, ex)
            mp.close()
    ix.close()
    _against_reference(record_property, ["-i", g["prefix"], "-f", f1, "-f2", f2, "-alg", "ksw2"], str(tmp_path / "gpu.ksw2.True.sam"), tmp_path)


def test_maximum_read_length_equals_oracle(api, tmp_path):
    """1000 bp reads (the longest the path takes) with enough errors that seeds are sparse and gap
    fragments run to hundreds of bases: the wide DP classes (up to 16 target columns per lane, traceback
    spilled to HBM) and the tier-1 capacities.  GPU SAM == oracle SAM, both algorithms."""
    from mapcaller_amd import synth
    g = synth.random_genome([500000, 300000], seed=909, n_repeats=6, repeat_len=900)
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, g)
    prefix = str(tmp_path / "g")
    api.Index.build(fa, prefix, 0)
    bases, _ = synth.simulate_reads(synth.mutate_genome(g, 4), 300, 1000, True, seed=6, skip_head=3000, frag_mean=2500, frag_sd=200, frag_min=2100,
                                    frag_max=3200, sub=0.03, ins=0.004, dele=0.004)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, bases, 0, 2); synth.write_fastq(f2, bases, 1, 2)
    ix = api.Index(prefix, device=0)
    for alg in ("ksw2", "nw"):
        mp = api.Mapper(ix, alg=alg, max_read_len=1000, max_batch_reads=600)
        out = str(tmp_path / f"gpu.{alg}.sam")
        st = mp.map_files(f1, f2, out)
        ora = str(tmp_path / f"ora.{alg}.sam")
        _oracle_sam(prefix, f1, f2, alg, ora)
        nd, ex = sam_diff(ora, out)
        assert nd == 0, (alg, ex)
        assert st["dp_jobs"] > 0
        mp.close()
    ix.close()


@pytest.mark.gpu
def test_host_without_torch_gets_the_same_records_over_the_boundary(api, golden, tmp_path):
    """python -m mapcaller_amd.boundary — libmcx.so through ctypes in a process that never loads torch, i.e. on the system's HIP runtime
    like a C/C++ host (bench.py's value_pcie_inclusive.system_runtime) — against this process (torch's runtime) on the same packed batches:
    the last batch's 32-byte records equal, field by field; and the child did run on the system's runtime."""
    import torch
    g = golden["var"]
    reads1 = [l for i, l in enumerate(open(g["r1"], "rb").read().split(b"\n")) if i % 4 == 1]
    reads2 = [l for i, l in enumerate(open(g["r2"], "rb").read().split(b"\n")) if i % 4 == 1]
    rlen = min(min(len(x) for x in reads1[:4000]), min(len(x) for x in reads2[:4000]))
    n_pairs, k = 1000, 4
    packed, host, meta_b = [], [], []
    for b in range(k):
        rows = np.stack([np.frombuffer(x[:rlen], dtype=np.uint8) for pno in range(b * n_pairs, (b + 1) * n_pairs) for x in (reads1[pno], reads2[pno])])
        w, l, o, n_odd, rw = api.pack_reads(torch.from_numpy(rows.copy()))
        host.append((w, l, o))
        packed.append((w.data_ptr(), rw, l.data_ptr(), o.data_ptr(), n_odd))
        w.numpy().tofile(tmp_path / f"batch{b}.words"); l.numpy().tofile(tmp_path / f"batch{b}.lens"); o.numpy().tofile(tmp_path / f"batch{b}.odd")
        meta_b.append({"row_words": rw, "n_odd": n_odd})
    n = 2 * n_pairs
    json.dump({"prefix": g["prefix"], "alg": "ksw2", "rlen": rlen, "reads": n, "steps": k, "full_sa": 1, "batches": meta_b, "dump": str(tmp_path / "recs.bin")},
              open(tmp_path / "meta.json", "w"))
    ix = api.Index(g["prefix"], device=0, full_sa=True)
    mp = api.Mapper(ix, alg="ksw2", max_batch_reads=n)
    outs = mp.stream_outputs(n, 3, 32)
    mp.map_stream_packed(packed[:3], n, True, outs, out32=True)  # (the child's warm-up sequence: the insert-size estimate is carried along)
    mp.map_stream_packed(packed, n, True, outs, out32=True)
    here = outs[(k - 1) % 3][0].numpy().view(api.ALN32_DTYPE).copy()
    mp.close(); ix.close()
    r = subprocess.run([sys.executable, "-m", "mapcaller_amd.boundary", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       cwd=os.path.join(os.path.dirname(__file__), ".."))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    there = np.fromfile(tmp_path / "recs.bin", dtype=api.ALN32_DTYPE)
    assert there.shape == here.shape
    for f in api.ALN32_DTYPE.names:  # (cigar_off is a place in the batch's CIGAR pool, taken with an atomic: the one field that may differ run to run)
        if f != "cigar_off":
            assert (there[f] == here[f]).all(), f
    assert line["mapped_frac_last_batch"] > 0.9
    v = subprocess.run([sys.executable, "-c", "import ctypes as C; L = C.CDLL('/opt/rocm/lib/libamdhip64.so.7'); v = C.c_int(); L.hipRuntimeGetVersion(C.byref(v)); print(v.value)"],
                       stdout=subprocess.PIPE, text=True, timeout=120, check=True).stdout  # (a process of its own: this one has torch's runtime loaded)
    assert line["hip_runtime_version"] == int(v)  # not the one torch's wheel carries


@pytest.mark.gpu
def test_bookkeeping_behind_the_batch_equals_bookkeeping_inside_the_call(api, golden, monkeypatch, capfd):
    """One shard: a batch's -vcf bookkeeping is queued behind it and runs under the next batch's kernels (the default), or runs inside the batch's call
    (MCX_NO_PROF_OVERLAP=1, and whenever HBM has no room for the second set of detail records).  The `var` reads in batches of 1000 both ways: the ten
    finalized planes and the tally records equal — and the first way did queue (MCX_TIMING says so), the second did not."""
    g = golden["var"]
    monkeypatch.setenv("MCX_TIMING", "1")
    got = []
    for inside in (False, True):
        if inside:
            monkeypatch.setenv("MCX_NO_PROF_OVERLAP", "1")
        ix = api.Index(g["prefix"], device=0, full_sa=True)
        mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1000)
        planes = api.planes_alloc(ix.genome_size, "cuda")
        mp.profile_attach(planes.data_ptr())
        capfd.readouterr()
        mp.map_files(g["r1"], g["r2"], None)
        mp.profile_finalize(planes.data_ptr())
        text = api.sparse_to_maps_text(mp.profile_sparse())
        err = capfd.readouterr().err
        assert ("queued behind the batch" in err) == (not inside)
        got.append((api.planes_view(planes, ix.genome_size).cpu().numpy().copy(), maps_canon(text)))
        mp.close(); ix.close()
    assert (got[0][0] == got[1][0]).all()
    assert got[0][1] == got[1][1]
