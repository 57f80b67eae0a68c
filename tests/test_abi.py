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



def test_pack_row_wide_form_equals_the_plain_one():
    """mcx_pack_row (the file front end's 2-bit rows, sixteen bases at a time with pext where the CPU has BMI2) against a plain Python
    restatement of include/mcx.h's rule — and against the library's own base-by-base form (MCX_PLAIN_PACK=1, a process of its own: the
    choice is made once).  Reads of every length up to 300 with lower-case letters, N and other bytes sprinkled in."""
    import subprocess, sys
    code = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %r)
lib = ctypes.CDLL(%r)
lib.mcx_pack_row.restype = ctypes.c_uint32
lib.mcx_pack_row.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
rng = np.random.default_rng(11)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
noise = np.frombuffer(b"acgtNnRY.*\n", dtype=np.uint8)
code = {65: 0, 67: 1, 71: 2, 84: 3}
out = []
for rlen in list(range(1, 301)) + [150] * 200:
    seq = alpha[rng.integers(0, 4, rlen)].copy()
    if rng.random() < 0.4:
        k = rng.integers(0, rlen, max(1, rlen // 20))
        seq[k] = noise[rng.integers(0, len(noise), len(k))]
    rw = (rlen + 15) // 16 + int(rng.integers(0, 2))
    row = np.full(rw + 1, 0xDEADBEEF, dtype=np.uint32)
    odd = np.zeros(400, dtype=np.uint64); n_odd = ctypes.c_uint32(0)
    got = lib.mcx_pack_row(seq.tobytes(), rlen, 7, row.ctypes.data, rw, odd.ctypes.data, 400, ctypes.byref(n_odd))
    want = np.zeros(rw, dtype=np.uint64); want_odd = []
    for i, b in enumerate(seq.tolist()):
        if b in code: want[i // 16] |= code[b] << (30 - 2 * (i %% 16))
        else: want_odd.append((7 << 32) | (i << 8) | b)
    assert got == len(want_odd) == n_odd.value, (rlen, got, len(want_odd))
    assert row[rw] == 0xDEADBEEF and (row[:rw] == want.astype(np.uint32)).all(), rlen
    assert odd[:got].tolist() == want_odd, rlen
    out.append(row[:rw].tobytes() + odd[:got].tobytes())
import hashlib
print(hashlib.sha1(b"".join(out)).hexdigest())
''' % (ROOT, _built("libmcx.so"))
    digests = []
    for env in ({}, {"MCX_PLAIN_PACK": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.strip())
    assert digests[0] == digests[1] and len(digests[0]) == 40


def test_host_cpus_is_what_the_process_is_given():
    """mcx_host_cpus: never more than the affinity mask, cut by a cgroup CPU-time share when there is one, overridden by MCX_HOST_CPUS."""
    import subprocess, sys
    code = "import ctypes; L = ctypes.CDLL(%r); L.mcx_host_cpus.restype = ctypes.c_uint32; print(L.mcx_host_cpus())" % _built("libmcx.so")
    run = lambda env: int(subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, stdout=subprocess.PIPE, text=True, timeout=120, check=True).stdout)
    n = run({})
    assert 1 <= n <= len(os.sched_getaffinity(0))
    share = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        share = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    if share:
        assert n <= max(1, int(share + 0.5))
    assert run({"MCX_HOST_CPUS": "3"}) == 3
    sys.path.insert(0, ROOT)
    import importlib
    bench_cpus = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.usable_cpus())" % ROOT], stdout=subprocess.PIPE, text=True, timeout=300, check=True).stdout
    assert int(bench_cpus) == n  # (bench.py runs the reference at -t <the same count>)
