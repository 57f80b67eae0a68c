"""The C-ABI libraries load and export every symbol include/*.h declares; the host-only entry points
(trajectory walk, in-process exchange, SAM merge) are exercised — no compute calls."""
import ctypes
import os
import re
import threading

import numpy as np

from conftest import ROOT


def _declared(header="mcx.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline[^{]*\{.*?\n\}", "", text, flags=re.S)  # (helpers defined in the header itself — mcx_aln_unpack — are no exports)
    names = set(re.findall(r"\b(mcx_[a-z0-9_]+)\s*\(", text))
    return sorted(n for n in names if not re.search(r"typedef\s+struct\s+" + n + r"\b", text))


def _built(name):
    import __graft_entry__ as g
    path = os.path.join(ROOT, "mapcaller_amd", name)
    if not os.path.exists(path):
        g.build()
    return path


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_built("libmcx.so"))
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n


def test_comm_library_exports_every_declared_symbol():
    """libmcx_comm.so (include/mcx_comm.h): the RCCL side, linked by mapcaller-mi355x -gpus N."""
    lib = ctypes.CDLL(_built("libmcx_comm.so"))
    names = _declared("mcx_comm.h")
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_lists_the_same_symbols():
    from mapcaller_amd import api
    assert sorted(api.SYMBOLS) == _declared()
    assert sorted(api.COMM_SYMBOLS) == _declared("mcx_comm.h")


def test_binding_fails_loudly_without_gpu_or_index():
    from mapcaller_amd import api
    import pytest
    with pytest.raises(api.McxError):
        api.Index("/nonexistent/prefix")


def test_avg_walk_is_the_references_feedback():
    """mcx_avg_walk against ReadMapping.cpp:462 / :538-539 written out: EstiDistance = (int)(avgDist * 1.5) per chunk,
    avgDist = (int)(TotalPairedDistance / iTotalPairedNum + .5) once more than 1000 pairs were seen."""
    from mapcaller_amd import api
    L = api.lib()
    rng = np.random.default_rng(5)
    ok = rng.integers(60, 101, size=40).astype(np.uint32)
    ds = (ok * rng.integers(380, 520, size=40)).astype(np.uint32)
    st = (ctypes.c_int64 * 3)(1000, 0, 0)
    est = np.zeros(40, dtype=np.int32)
    L.mcx_avg_walk(st, ok.ctypes.data, ds.ctypes.data, 40, est.ctypes.data)
    avg, tp, td, want = 1000, 0, 0, []
    for k in range(40):
        want.append(int(avg * 1.5))
        tp += int(ok[k]); td += int(ds[k])
        if tp > 1000:
            avg = int(1.0 * td / tp + .5)
    assert est.tolist() == want and list(st) == [avg, tp, td]
    # two shards walking halves one after the other end where one walk over everything ends
    st2 = (ctypes.c_int64 * 3)(1000, 0, 0)
    L.mcx_avg_walk(st2, ok.ctypes.data, ds.ctypes.data, 17, None)
    L.mcx_avg_walk(st2, ok[17:].ctypes.data, ds[17:].ctypes.data, 23, None)
    assert list(st2) == list(st)


def test_local_exchange_gathers_in_rank_order():
    """mcx_exchange_local: the rendezvous the host threads of mapcaller-mi355x -gpus N share."""
    from mapcaller_amd import api
    L = api.lib()
    n = 4
    links = (api.Exchange * n)()
    assert L.mcx_exchange_local(n, links) == 0
    out = [None] * n

    def work(r):
        got = []
        for rnd in range(50):
            send = np.full(8 + rnd, (16 * rnd + r) % 256, dtype=np.uint8)
            recv = np.zeros(n * send.size, dtype=np.uint8)
            assert links[r].allgather(links[r].user, send.ctypes.data, recv.ctypes.data, send.size) == 0
            got.append(recv.reshape(n, -1)[:, 0].tolist())
        out[r] = got

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r in range(n):
        assert out[r] == [[(16 * rnd + q) % 256 for q in range(n)] for rnd in range(50)]
    L.mcx_exchange_local_free(links)

