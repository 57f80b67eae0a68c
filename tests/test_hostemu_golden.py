"""Host logic: the product's device headers (mcx_fm.h, mcx_glue.h) compiled for the host by
tests/hostemu, run with the product's stage order, capacity tiers and avgDist replay, against the
reference's golden SAM.  (The wavefront DP kernels themselves are covered by the gpu tests.)"""
import ctypes

import pytest

from conftest import SETS, VCF_CASES, VCF_RUNS, VcfOpts, sam_diff, vcf_alg, vcf_body


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
