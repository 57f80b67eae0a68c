"""The oracle (CPU restatement) against the reference's golden vectors — this is what pins it."""
import ctypes
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from conftest import GOLD, SETS, VCF_CASES, VCF_RUNS, VcfOpts, maps_canon, sam_diff, vcf_alg, vcf_body


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", list(SETS))
def test_oracle_sam_equals_reference(oracle_lib, golden, tmp_path, name, alg):
    g = golden[name]
    ix = oracle_lib.mcxo_index_load(g["prefix"].encode())
    assert ix
    out = str(tmp_path / "o.sam")
    st = (ctypes.c_int64 * 8)()
    n = oracle_lib.mcxo_map_files(ix, g["r1"].encode(), (g["r2"] or "").encode(), 0 if alg == "nw" else 1, out.encode(), 1, st)
    oracle_lib.mcxo_index_free(ix)
    assert n > 0
    nd, ex = sam_diff(g["sam"][alg], out)
    assert nd == 0, ex


def test_oracle_bwt_search_vectors(oracle_lib, golden):
    q = json.load(open(os.path.join(GOLD, "func", "bwt_search.json")))
    ix = oracle_lib.mcxo_index_load(golden["toy"]["prefix"].encode())
    ln, fr = ctypes.c_int(), ctypes.c_int()
    loc = (ctypes.c_uint64 * 50)()
    for rec in q:
        codes = bytes("ACGTN".index(c) for c in rec["seq"])
        oracle_lib.mcxo_bwt_search(ix, codes, rec["start"], len(codes), ctypes.byref(ln), ctypes.byref(fr), loc)
        assert (ln.value, fr.value) == (rec["len"], rec["freq"])
        assert [int(loc[i]) for i in range(fr.value)] == rec["loc"]
    oracle_lib.mcxo_index_free(ix)


def test_oracle_dp_vectors(oracle_lib):
    cases = json.load(open(os.path.join(GOLD, "func", "dp.json")))
    for rec in cases:
        q, t = rec["q"].encode(), rec["t"].encode()
        cap = len(q) + len(t) + 8
        o1, o2 = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
        oracle_lib.mcxo_nw(q, len(q), t, len(t), o1, o2, cap)
        assert [o1.value.decode(), o2.value.decode()] == rec["nw"]
        oracle_lib.mcxo_ksw2(q, len(q), t, len(t), o1, o2, cap)
        assert [o1.value.decode(), o2.value.decode()] == rec["ksw2"]
        sc = ctypes.c_int()
        ops = ctypes.create_string_buffer(cap)
        qc = bytes("ACGTN".index(c) for c in rec["q"])
        tc = bytes("ACGTN".index(c) for c in rec["t"])
        oracle_lib.mcxo_ksw2_extz(qc, len(qc), tc, len(tc), ctypes.byref(sc), ops, cap)
        assert sc.value == rec["ksw2_score"]
        assert ops.value.decode() == rec["ksw2_ops_rev"]


@pytest.mark.parametrize("name", list(SETS))
def test_oracle_profile_equals_reference(oracle_lib, golden, tmp_path, name):
    """UpdateProfile / UpdateMultiHitCount restated: the per-position counters and the sparse maps
    against what the reference's Mapping() left behind (oracle/_ref/mcref_tool P)."""
    g = golden[name]
    alg, prof, maps = g["prof"]
    ix = oracle_lib.mcxo_index_load(g["prefix"].encode())
    out = str(tmp_path / "p")
    n = oracle_lib.mcxo_map_files_profile(ix, g["r1"].encode(), (g["r2"] or "").encode(), 0 if alg == "nw" else 1, out.encode())
    oracle_lib.mcxo_index_free(ix)
    assert n > 0
    assert open(out + ".prof", "rb").read() == open(prof, "rb").read()
    assert maps_canon(open(out + ".maps", encoding="latin-1").read()) == maps_canon(open(maps, encoding="latin-1").read())


@pytest.mark.parametrize("name,tag", VCF_CASES)
def test_oracle_vcf_equals_reference(oracle_lib, golden, tmp_path, name, tag):
    """VariantCalling() restated (reference src/VariantCalling.cpp): the VCF of `MapCaller -vcf -t 1`
    line for line — SNVs, indels, gaps, duplicated regions, <INV>/<TNL> break points, and the
    -gvcf / -monomorphic / -filter / -ploidy / -somatic / threshold switches."""
    g = golden[name]
    ix = oracle_lib.mcxo_index_load(g["prefix"].encode())
    out = str(tmp_path / "o.vcf")
    opts = VcfOpts(VCF_RUNS[tag][1])
    n = oracle_lib.mcxo_map_files_vcf(ix, g["r1"].encode(), (g["r2"] or "").encode(), 0 if vcf_alg(name, tag) == "nw" else 1, out.encode(), opts.ref)
    oracle_lib.mcxo_index_free(ix)
    assert n > 0
    assert vcf_body(out) == vcf_body(g["vcf"][tag])


def test_oracle_input_side_cases(oracle_lib, io_golden, tmp_path):
    """GetData.cpp semantics restated: interleaved -p input whose odd 199-read tail chunk is mapped as
    single reads, headers cut at ' ' or '/', multi-line FASTA."""
    g = io_golden
    ix = oracle_lib.mcxo_index_load(g["prefix"].encode())
    out = str(tmp_path / "il.sam")
    assert oracle_lib.mcxo_map_files_interleaved(ix, g["il.fq"].encode(), 1, out.encode(), None) == 2999
    nd, ex = sam_diff(g["ref.il.sam"], out, mask_se_reverse_qual=True)
    assert nd == 0, ex
    out = str(tmp_path / "ml.sam")
    assert oracle_lib.mcxo_map_files(ix, g["ml.fa"].encode(), b"", 0, out.encode(), 1, None) == 400
    nd, ex = sam_diff(g["ref.ml.sam"], out)
    assert nd == 0, ex
    oracle_lib.mcxo_index_free(ix)


def test_ref_tool_whole_run_equals_the_plain_binary(golden, tmp_path):
    """mcref_tool's request R — the reference's own main() once, -sam and -vcf, with the accumulated profile and maps dumped between Mapping() and
    VariantCalling() through a hook at main.cpp:380 (oracle/Makefile: -DVariantCalling=mcref_vc_hook for main_lib.o) — against the plain binary's SAM
    and VCF and request Q's dump of the same reads: the one reference pass the full-size config-4 test compares the GPU with is the reference."""
    import gzip
    import subprocess
    ref_bin, ref_tool = os.path.join(ROOT, "oracle", "_ref", "MapCaller"), os.path.join(ROOT, "oracle", "_ref", "mcref_tool")
    if not (os.path.exists(ref_bin) and os.path.exists(ref_tool)):
        pytest.skip("oracle/_ref is not built here")
    g = golden["var"]
    a_sam, a_vcf, b_sam, b_vcf = (str(tmp_path / n) for n in ("a.sam", "a.vcf", "b.sam", "b.vcf"))
    subprocess.run([ref_bin, "-i", g["prefix"], "-f", g["r1"], "-f2", g["r2"], "-alg", "ksw2", "-sam", a_sam, "-vcf", a_vcf, "-t", "1", "-log", os.devnull],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([ref_tool], input=f"R ksw2 {tmp_path}/r {g['prefix']} {b_sam} {b_vcf} {g['r1']} {g['r2']}\n", text=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    assert r.stdout.strip().split("\n")[-1] == "ok", r.stdout[-300:]
    r = subprocess.run([ref_tool], input=f"L {g['prefix']}\nQ ksw2 {tmp_path}/q {g['r1']} {g['r2']}\n", text=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    assert r.stdout.strip().split("\n")[-1] == "ok", r.stdout[-300:]
    assert open(a_sam, "rb").read() == open(b_sam, "rb").read()
    body = lambda p: [l for l in open(p) if not l.startswith("##")]
    assert body(a_vcf) == body(b_vcf) and len(body(a_vcf)) > 10
    for ext in (".prof.nz", ".maps"):
        assert open(f"{tmp_path}/r{ext}", "rb").read() == open(f"{tmp_path}/q{ext}", "rb").read()
    assert os.path.getsize(f"{tmp_path}/r.prof.nz") > 100_000
