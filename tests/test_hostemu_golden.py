"""Host logic: the product's device headers (mcx_fm.h, mcx_glue.h, mcx_dp_lane.h) compiled for the host by
tests/hostemu, run with the product's stage order, capacity tiers and avgDist replay, against the
reference's golden SAM and function-level vectors.  The one-problem-per-lane DP is plain per-lane code and runs
here as it does on the GPU; the wavefront DP kernels (wave shuffles, LDS) are covered by the gpu tests."""
import ctypes
import json
import os

import pytest

from conftest import GOLD, SETS, VCF_CASES, VCF_RUNS, VcfOpts, sam_diff, vcf_alg, vcf_body


def _run(lib, g, alg, out, batch=1 << 20, tier0=None, rlen_max=256):
    st = (ctypes.c_int64 * 12)()
    t0 = (ctypes.c_int * 5)(*tier0) if tier0 else None
    n = lib.hostemu_map_files(g["prefix"].encode(), g["r1"].encode(), (g["r2"] or "").encode(), 0 if alg == "nw" else 1,
                              out.encode(), batch, t0, rlen_max, st)
    return n, list(st)


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", list(SETS))
def test_device_glue_on_host_equals_reference(hostemu_lib, golden, tmp_path, name, alg):
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden[name], alg, out)
    assert n > 0
    nd, ex = sam_diff(golden[name]["sam"][alg], out)
    assert nd == 0, ex


@pytest.mark.parametrize("K", [8, 16])
@pytest.mark.parametrize("alg", ["nw", "ksw2"])
def test_lane_dp_equals_reference_vectors(hostemu_lib, alg, K):
    """The one-problem-per-lane DP (strips of K columns, packed traceback flags, row-major sweep) on the reference's own
    nw_alignment / ksw2_alignment vectors (tests/golden/func/dp.json, made by oracle/_ref/mcref_tool): the gapped strings —
    hence every traceback flag the walk visits — and for ksw2 the reversed operation string of ksw_backtrack.  Targets with
    an N are left out: the pipeline's targets come from the 2-bit genome."""
    cases = json.load(open(os.path.join(GOLD, "func", "dp.json")))
    hostemu_lib.hostemu_lane_dp.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    done = 0
    for c in cases:
        q, t = c["q"].encode(), c["t"].encode()
        buf = ctypes.create_string_buffer(len(q) + len(t) + 2)
        sc = ctypes.c_int()
        L = hostemu_lib.hostemu_lane_dp(1 if alg == "nw" else 0, q, len(q), t, len(t), K, buf, ctypes.byref(sc))
        if L < 0:
            continue
        ops = buf.value.decode()
        a1, a2, qi, ti = [], [], 0, 0
        for o in ops:  # what the column string says about the two gapped strings
            if o == "M":
                a1.append(c["q"][qi]); a2.append(c["t"][ti]); qi += 1; ti += 1
            elif o == "I":
                a1.append(c["q"][qi]); a2.append("-"); qi += 1
            else:
                a1.append("-"); a2.append(c["t"][ti]); ti += 1
        assert ["".join(a1), "".join(a2)] == c[alg], (c["q"], c["t"], ops)
        if alg == "ksw2":
            assert ops[::-1] == c["ksw2_ops_rev"]
        done += 1
    assert done > 550


def _gapped(c, ops):
    a1, a2, qi, ti = [], [], 0, 0
    for o in ops:  # what the column string says about the two gapped strings
        if o == "M":
            a1.append(c["q"][qi]); a2.append(c["t"][ti]); qi += 1; ti += 1
        elif o == "I":
            a1.append(c["q"][qi]); a2.append("-"); qi += 1
        else:
            a1.append("-"); a2.append(c["t"][ti]); ti += 1
    return ["".join(a1), "".join(a2)]


@pytest.mark.parametrize("K", [8, 16])
@pytest.mark.parametrize("alg", ["nw", "ksw2"])
def test_two_problems_per_lane_dp_equals_the_reference_vectors(hostemu_lib, alg, K):
    """The two-problems-per-lane DP (mcx_dp_lane2.h: 16-bit halves, the arithmetic of k_dp_lane2 with the packed instructions spelt out in C) on the
    reference's function-level vectors: every vector shares its lane once with its neighbour of the list and once with a vector of another
    shape (a third of the list away), once in the low half and once in the high one — the gapped strings of nw_alignment / ksw2_alignment, for ksw2
    the reversed operation string of ksw_backtrack, for nw the final score against the one-problem-per-lane sweep's."""
    cases = [c for c in json.load(open(os.path.join(GOLD, "func", "dp.json"))) if set(c["t"]) <= set("ACGT")]
    L = hostemu_lib
    L.hostemu_lane_dp2.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.hostemu_lane_dp.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    use_nw = 1 if alg == "nw" else 0
    n, done, shapes = len(cases), 0, set()
    for i, a in enumerate(cases):
        for b in (cases[(i + 1) % n], cases[(i + n // 3) % n]):
            for lo, hi in ((a, b), (b, a)):
                qa, ta, qb, tb = lo["q"].encode(), lo["t"].encode(), hi["q"].encode(), hi["t"].encode()
                oa, ob = ctypes.create_string_buffer(len(qa) + len(ta) + 2), ctypes.create_string_buffer(len(qb) + len(tb) + 2)
                sc, ln = (ctypes.c_int * 2)(), (ctypes.c_int * 2)()
                assert L.hostemu_lane_dp2(use_nw, qa, len(qa), ta, len(ta), qb, len(qb), tb, len(tb), K, oa, ob, sc, ln) == 0
                for c, o, k in ((lo, oa, 0), (hi, ob, 1)):
                    ops = o.value.decode()
                    assert len(ops) == ln[k]
                    assert _gapped(c, ops) == c[alg], (k, c["q"], c["t"], ops)
                    if alg == "ksw2":
                        assert ops[::-1] == c["ksw2_ops_rev"]
                    else:
                        buf, s1 = ctypes.create_string_buffer(len(c["q"]) + len(c["t"]) + 2), ctypes.c_int()
                        L.hostemu_lane_dp(1, c["q"].encode(), len(c["q"]), c["t"].encode(), len(c["t"]), K, buf, ctypes.byref(s1))
                        assert s1.value == sc[k]
                shapes.add((len(qa) > len(qb), len(ta) > len(tb)))
                done += 1
    assert done > 2000 and len(shapes) == 4  # (every combination of longer / shorter query and target in either half)


@pytest.mark.parametrize("K", [8, 16])
@pytest.mark.parametrize("alg", ["nw", "ksw2"])
def test_two_problems_per_lane_dp_equals_one_per_lane_on_random_shapes(hostemu_lib, alg, K):
    """Seeded random problems — a query and a target that descend from one sequence by substitutions, insertions and deletions, 1 to 300 by 1 to 256 bases
    (one to sixteen strips, walks that cross strips and window edges at every offset), now and then an N in the query — through the two-problems-per-lane
    DP, each pair of neighbours sharing a lane both ways round: the column strings and nw's score equal the one-problem-per-lane form's, which the
    reference's vectors hold (above)."""
    import random
    rng = random.Random(20261004 + K + (0 if alg == "nw" else 1))
    L = hostemu_lib
    L.hostemu_lane_dp2.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.hostemu_lane_dp.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    use_nw = 1 if alg == "nw" else 0
    max_t = 8 * K if K == 8 else 256

    def problem():
        n = rng.randint(1, max_t)
        t = "".join(rng.choice("ACGT") for _ in range(n))
        rate = rng.choice([0.0, 0.02, 0.05, 0.15])
        q = []
        for ch in t:
            r = rng.random()
            if r < rate / 3: continue                                   # deletion
            if r < 2 * rate / 3: q.append(rng.choice("ACGT"))            # insertion before
            q.append(rng.choice("ACGT") if rng.random() < rate else ch)  # substitution
        if rng.random() < 0.2: q = q[: rng.randint(1, max(1, len(q)))]   # ragged: a short query against a long target
        if rng.random() < 0.2: q = q + [rng.choice("ACGT") for _ in range(rng.randint(1, 40))]
        q = q[:300] or ["A"]
        if rng.random() < 0.1: q[rng.randrange(len(q))] = "N"
        return "".join(q), t

    def one(q, t):
        buf, sc = ctypes.create_string_buffer(len(q) + len(t) + 2), ctypes.c_int()
        ln = L.hostemu_lane_dp(use_nw, q.encode(), len(q), t.encode(), len(t), K, buf, ctypes.byref(sc))
        return buf.value.decode(), sc.value, ln

    probs = [problem() for _ in range(1500)]
    ref = [one(q, t) for q, t in probs]
    for i in range(len(probs)):
        j = (i + 1) % len(probs)
        (qa, ta), (qb, tb) = probs[i], probs[j]
        oa, ob = ctypes.create_string_buffer(len(qa) + len(ta) + 2), ctypes.create_string_buffer(len(qb) + len(tb) + 2)
        sc, ln = (ctypes.c_int * 2)(), (ctypes.c_int * 2)()
        assert L.hostemu_lane_dp2(use_nw, qa.encode(), len(qa), ta.encode(), len(ta), qb.encode(), len(qb), tb.encode(), len(tb), K, oa, ob, sc, ln) == 0
        for k, (o, r) in enumerate(((oa, ref[i]), (ob, ref[j]))):
            assert o.value.decode() == r[0], (k, probs[i if k == 0 else j])
            assert ln[k] == r[2]
            if alg == "nw":
                assert sc[k] == r[1]


@pytest.mark.parametrize("switch", ["MCX_EMU_NO_CODES", "MCX_EMU_ORACLE_DP", "MCX_EMU_DP_X1"])
@pytest.mark.parametrize("alg", ["nw", "ksw2"])
def test_lane_dp_inputs_and_the_scalar_dp_agree(hostemu_lib, golden, tmp_path, monkeypatch, alg, switch):
    """The `var` set (indel-rich donors: hundreds of DP problems, both strands) with the lane DP reading its query from the
    ASCII bases instead of the 2-bit words, and with the oracle's scalar DP in its place: the same SAM as the reference's."""
    monkeypatch.setenv(switch, "1")
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden["var"], alg, out)
    assert st[6] > 100  # DP problems
    nd, ex = sam_diff(golden["var"]["sam"][alg], out)
    assert nd == 0, ex


@pytest.mark.parametrize("name", list(SETS))
def test_two_base_steps_of_the_walk_change_nothing(hostemu_lib, golden, tmp_path, monkeypatch, name):
    """The seeding walk over the pair records (mcx_fm.h PairSlot: two bases per step, the full suffix array beside them) gives the
    reference's SAM, with fewer index records fetched than the single steps take."""
    out = str(tmp_path / "e.sam")
    _, st1 = _run(hostemu_lib, golden[name], "ksw2", out)
    monkeypatch.setenv("MCX_EMU_RANK2", "1")
    n, st2 = _run(hostemu_lib, golden[name], "ksw2", out)
    assert n > 0
    nd, ex = sam_diff(golden[name]["sam"]["ksw2"], out)
    assert nd == 0, ex
    assert st2[3] == st1[3]          # the searches end where they ended: the same bases consumed
    assert 0 < st2[8] < st1[8]       # in fewer fetches


@pytest.mark.parametrize("name", list(SETS))
def test_pair_records_equal_two_single_steps(hostemu_lib, golden, name):
    """mcx_fm.h's self-check of the pair records (the one MCX_RANK2_CHECK runs on the device): the bi-intervals of random strings
    of 1..12 bases — every width, with and without the primary row and the lone suffix inside — extended by two random bases
    through one pair step and through two single steps in the .bwt blocks."""
    hostemu_lib.hostemu_pair_check.restype = ctypes.c_int64
    hostemu_lib.hostemu_pair_check.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    assert hostemu_lib.hostemu_pair_check(golden[name]["prefix"].encode(), 400000) == 0


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", ["var", "long", "toy"])
def test_straight_line_pairs_with_their_own_dp_problems(hostemu_lib, golden, tmp_path, monkeypatch, name, alg):
    """mcx_simple.h's three passes (collect, solve, replay): pairs whose only obstacle is a small gapped extension — between two seeds or
    at a read end, where the outer gap columns are stripped, soft-clipped or the whole end dropped — take the straight-line path; the
    SAM is the reference's with them, without the DP problems (MCX_EMU_SIMPLE_NO_DP) and with every pair on the general path."""
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden[name], alg, out)
    nd, ex = sam_diff(golden[name]["sam"][alg], out)
    assert nd == 0, ex
    monkeypatch.setenv("MCX_EMU_SIMPLE_NO_DP", "1")
    _, st_no_dp = _run(hostemu_lib, golden[name], alg, out)
    nd, ex = sam_diff(golden[name]["sam"][alg], out)
    assert nd == 0, ex
    monkeypatch.setenv("MCX_EMU_NO_SIMPLE", "1")
    _, st_general = _run(hostemu_lib, golden[name], alg, out)
    nd, ex = sam_diff(golden[name]["sam"][alg], out)
    assert nd == 0, ex
    assert st[11] > st_no_dp[11] > 0 and st_general[11] == 0, (st[11], st_no_dp[11], st_general[11])  # pairs the path took
    assert st[6] < st_no_dp[6] <= st_general[6]                                                        # DP problems left to the general path


@pytest.mark.parametrize("alg", ["nw", "ksw2"])
@pytest.mark.parametrize("name", list(SETS))
def test_straight_line_pairs_write_the_general_paths_detail_records(hostemu_lib, golden, tmp_path, monkeypatch, name, alg):
    """With the -vcf bookkeeping on, the straight-line path writes the per-read detail records itself (mcx_simple.h SimpleDetail: what
    k_simple<.., DETAIL> runs per lane) instead of leaving its pairs to the finish stage.  Every pair the path takes goes through the
    general path as well here, and the two records of each of its reads are compared: type, orientation, every fragment (kind, read and
    genome span) in order, the column strings of DP fragments (trimmed and dropped ends included), and in read 1's record the
    discordant-pair fields of pair_stats (ReadMapping.cpp:486-521).  What UpdateProfile reads is then the same either way."""
    monkeypatch.setenv("MCX_EMU_DETAIL_CHECK", "1")
    cnt = (ctypes.c_int64 * 2)()
    hostemu_lib.hostemu_detail_counts(cnt)  # (start over)
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden[name], alg, out)
    nd, ex = sam_diff(golden[name]["sam"][alg], out)
    assert nd == 0, ex
    hostemu_lib.hostemu_detail_counts(cnt)
    assert cnt[0] >= st[11] > 0 and cnt[1] == 0, (cnt[0], cnt[1], st[11])


@pytest.mark.parametrize("name", ["toy", "mc"])
def test_comparison_phase_forms_agree(hostemu_lib, golden, name):
    """The seeding walk's comparison phase 16 bases per fetch, 64 per fetch and 64-then-128 with the shared chunk kept: the same end of the
    seed for reads cut out of the text on both strands, across the strand boundary and the text's end, with a changed base or an N."""
    hostemu_lib.hostemu_compare_check.restype = ctypes.c_int64
    hostemu_lib.hostemu_compare_check.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_uint64]
    assert hostemu_lib.hostemu_compare_check(golden[name]["prefix"].encode(), 200000, 2463534242) == 0


def test_pairing_sweep_equals_the_scan_of_all_pairs(hostemu_lib):
    """pair_by_distance sweeps read 2's candidates in PosDiff order; on random candidate lists (ties, dead candidates, lists long enough
    for RemoveRedundantAlnCan) it pairs the same candidates and reports the same interval of estimates as the scan of all n1 x n2 pairs."""
    hostemu_lib.hostemu_pairing_check.restype = ctypes.c_int64
    hostemu_lib.hostemu_pairing_check.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    assert hostemu_lib.hostemu_pairing_check(300000, 88172645463325252) == 0


def test_small_batches_follow_the_avgdist_trajectory(hostemu_lib, golden, tmp_path):
    # 600-read batches: the insert-size estimate crosses batch boundaries
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden["mc"], "ksw2", out, batch=600)
    nd, ex = sam_diff(golden["mc"]["sam"]["ksw2"], out)
    assert nd == 0, ex


def test_capacity_tiers_do_not_change_results(hostemu_lib, golden, tmp_path):
    # absurdly small tier-0 capacities push most pairs through tier 1
    out = str(tmp_path / "e.sam")
    n, st = _run(hostemu_lib, golden["mc"], "nw", out, tier0=[4, 2, 12, 64, 2])
    assert st[9] > 100  # tier-1 pairs
    nd, ex = sam_diff(golden["mc"]["sam"]["nw"], out)
    assert nd == 0, ex


@pytest.mark.parametrize("grain", [None, "7"])
@pytest.mark.parametrize("name,tag", [c for c in VCF_CASES if c[1] not in ("opts", "nw")])
def test_variant_host_logic_equals_reference(hostemu_variants_lib, oracle_lib, golden, tmp_path, monkeypatch, name, tag, grain):
    """The product's variant-calling host logic (mcx_variants_host.h: indel calls, run pairing, gVCF
    blocks, break points, filters, VCF text) over a CPU stand-in for the dense kernels, fed with the
    reference's own profile dump: the golden VCF line for line.  (The runs whose profile needs other
    -dup / -maxclip / -alg values have no dump and are covered on the GPU.)  grain "7": the passes over the records
    are cut into stretches of seven records for the host threads (MCX_VC_GRAIN), so that runs, strings and sorted
    lists cross stretch borders on these small inputs as they do at genome scale."""
    if grain:
        monkeypatch.setenv("MCX_VC_GRAIN", grain)
    g = golden[name]
    alg, prof, maps = g["prof"]
    assert alg == vcf_alg(name, tag)
    ix = oracle_lib.mcxo_index_load(g["prefix"].encode())
    tot = (ctypes.c_int64 * 3)()
    assert oracle_lib.mcxo_pair_totals(ix, g["r1"].encode(), (g["r2"] or "").encode(), 0 if alg == "nw" else 1, tot) > 0
    oracle_lib.mcxo_index_free(ix)
    out = str(tmp_path / "o.vcf")
    opts = VcfOpts(VCF_RUNS[tag][1])
    rc = hostemu_variants_lib.hostemu_call_variants(g["prefix"].encode(), prof.encode(), maps.encode(), tot[0], tot[1], tot[2], opts.ref, out.encode())
    assert rc == 0, hostemu_variants_lib.hostemu_vc_error()
    assert vcf_body(out) == vcf_body(g["vcf"][tag])
