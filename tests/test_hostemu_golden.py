"""Host logic: the product's device headers (mcx_fm.h, mcx_glue.h) compiled for the host by
tests/hostemu, run with the product's stage order, capacity tiers and avgDist replay, against the
reference's golden SAM.  (The wavefront DP kernels themselves are covered by the gpu tests.)"""
import ctypes

import pytest

from conftest import SETS, sam_diff


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
