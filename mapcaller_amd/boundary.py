"""The host-buffer leg of bench.py in a process of its own — one that never loads torch.

    python -m mapcaller_amd.boundary <dir>

Why: a process that has imported torch runs on the HIP runtime torch's wheel carries (7.0.2 in this image), and that runtime puts the
copies of BOTH directions on SDMA engine 0 (AMD_LOG_LEVEL=4: "HSA Copy copy_engine=0x1" for either; 320 MB in + 256 MB out take
10.6 ms together, 5.9 + 4.7 alone: scripts/probe/d2h_probe.py); the system's runtime (ROCm 7.2, which libmcx.so and the native CLI
link) asks the driver for the engines it recommends per direction and the two copies overlap (5.96 ms together:
scripts/probe/d2h_probe.hip).  A C/C++ host — the CLI, the reference with the binding of INTEGRATION.md — is in the second
situation; this module measures that one: libmcx.so through ctypes, numpy, nothing else.

<dir> holds what bench.py wrote: genome.u8 (codes 0..3), meta.json ({"chr_lens", "alg", "rlen", "reads", "steps", "full_sa",
"batches": [{"row_words", "n_odd"}]}), batch<i>.words / .lens / .odd.  Prints one JSON line.  (Tests: "prefix" instead of the genome — an
index on disk —, and "dump": a file that takes the last batch's records as they arrived, 32 bytes each.)"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np


def main(d: str) -> None:
    os.environ["MCX_NO_TORCH"] = "1"
    from mapcaller_amd import api
    api.lib()
    assert "torch" not in sys.modules
    hip = C.CDLL("libamdhip64.so.7")  # (the one libmcx.so brought in)
    for f in ("hipMalloc", "hipHostMalloc"):
        getattr(hip, f).argtypes = [C.POINTER(C.c_void_p), C.c_size_t] + ([C.c_uint] if f == "hipHostMalloc" else [])
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipRuntimeGetVersion.argtypes = [C.POINTER(C.c_int)]

    def ok(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: hip error {rc}")

    def pinned(a: np.ndarray):
        p = C.c_void_p()
        ok(hip.hipHostMalloc(C.byref(p), max(a.nbytes, 16), 0), "hipHostMalloc")
        C.memmove(p, a.ctypes.data, a.nbytes)
        return p

    meta = json.load(open(os.path.join(d, "meta.json")))
    ver = C.c_int()
    hip.hipRuntimeGetVersion(C.byref(ver))
    if meta.get("prefix"):
        index = api.Index(meta["prefix"], device=0, full_sa=bool(meta["full_sa"]))
        index.build_seconds = 0.0
    else:
        codes = np.fromfile(os.path.join(d, "genome.u8"), dtype=np.uint8)
        dp = C.c_void_p()
        ok(hip.hipMalloc(C.byref(dp), codes.nbytes), "hipMalloc")
        ok(hip.hipMemcpy(dp, codes.ctypes.data, codes.nbytes, 1), "hipMemcpy")
        index = api.Index.from_codes(dp.value, meta["chr_lens"], device=0, full_sa=meta["full_sa"])
        ok(hip.hipFree(dp), "hipFree")
        del codes
    n = meta["reads"]
    mapper = api.Mapper(index, alg=meta["alg"], max_read_len=max(256, meta["rlen"]), max_batch_reads=n)
    packed = []
    for i, b in enumerate(meta["batches"]):
        w = pinned(np.fromfile(os.path.join(d, f"batch{i}.words"), dtype=np.int32))
        ln = pinned(np.fromfile(os.path.join(d, f"batch{i}.lens"), dtype=np.int32))
        od = pinned(np.fromfile(os.path.join(d, f"batch{i}.odd"), dtype=np.int64))
        packed.append((w.value, b["row_words"], ln.value, od.value, b["n_odd"]))
    k = meta["steps"]
    packed = [packed[i % len(packed)] for i in range(k)]

    class Buf:  # what map_stream_packed asks of an output buffer
        def __init__(self, nbytes):
            self.p = C.c_void_p()
            ok(hip.hipHostMalloc(C.byref(self.p), nbytes, 0), "hipHostMalloc")

        def data_ptr(self):
            return self.p.value

    outs = [(Buf(n * 32), Buf(api.cigar_pool_words(n) * 4)) for _ in range(3)]
    b0 = mapper.map_stream_packed(packed[:3], n, True, outs, out32=True)
    ok(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
    t0 = time.perf_counter()
    b1 = mapper.map_stream_packed(packed, n, True, outs, out32=True)
    dt = time.perf_counter() - t0
    recs = np.ctypeslib.as_array(C.cast(outs[(k - 1) % 3][0].p, C.POINTER(C.c_uint8)), shape=(n * 32,)).view(api.ALN32_DTYPE)
    if meta.get("dump"):
        recs.tofile(meta["dump"])
    print(json.dumps({"value": round(k * n / dt, 1), "unit": "reads/s", "steps": k, "ms_per_step": round(1000 * dt / k, 3),
                      "h2d_bytes_per_read": round((b1[0] - b0[0]) / (k * n), 1), "d2h_bytes_per_read": round((b1[1] - b0[1]) / (k * n), 1),
                      "hip_runtime_version": ver.value, "index_build_s": round(index.build_seconds, 2),
                      "mapped_frac_last_batch": round(float((api.aln32_unpack(recs)["chr"] >= 0).mean()), 4)}))
    mapper.close()


if __name__ == "__main__":
    main(sys.argv[1])
