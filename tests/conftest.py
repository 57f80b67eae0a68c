"""Shared fixtures.  `-m "not gpu"` covers the oracle against the golden vectors, the host logic
(product device headers compiled for the host by tests/hostemu) and the C-ABI surface;
`-m gpu` are the parity tests proper: they call the HIP path through libmcx.so."""
import gzip
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
SETS = {"toy": True, "mc": True, "se": False, "long": True}  # name -> paired


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through libmcx.so)")


def _make(directory, target=None):
    cmd = ["make", "-C", directory] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


@pytest.fixture(scope="session")
def oracle_lib():
    import ctypes
    path = os.path.join(ROOT, "oracle", "libmcx_oracle.so")
    if not os.path.exists(path):
        _make(os.path.join(ROOT, "oracle"))
    L = ctypes.CDLL(path)
    L.mcxo_index_load.restype = ctypes.c_void_p
    L.mcxo_index_load.argtypes = [ctypes.c_char_p]
    L.mcxo_index_free.argtypes = [ctypes.c_void_p]
    L.mcxo_map_files.restype = ctypes.c_int64
    L.mcxo_map_files.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                 ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    L.mcxo_map_files_profile.restype = ctypes.c_int64
    L.mcxo_map_files_profile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.mcxo_bwt_search.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
    for f in (L.mcxo_nw, L.mcxo_ksw2):
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    L.mcxo_ksw2_extz.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                 ctypes.c_char_p, ctypes.c_int]
    return L


@pytest.fixture(scope="session")
def hostemu_lib(oracle_lib):
    import ctypes
    d = os.path.join(ROOT, "tests", "hostemu")
    _make(d)
    L = ctypes.CDLL(os.path.join(d, "libhostemu.so"))
    L.hostemu_map_files.restype = ctypes.c_int64
    L.hostemu_map_files.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                                           ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    return L


@pytest.fixture(scope="session")
def golden(tmp_path_factory):
    """Unpacks tests/golden/<set>/ into a temp dir: returns name -> dict(prefix, r1, r2, sam[alg])."""
    out = {}
    base = tmp_path_factory.mktemp("golden")
    for name, paired in SETS.items():
        src = os.path.join(GOLD, name)
        dst = base / name
        dst.mkdir()
        for ext in ("bwt", "sa", "pac", "ann", "amb"):
            shutil.copy(os.path.join(src, f"idx.{ext}"), dst / f"idx.{ext}")
        info = {"prefix": str(dst / "idx"), "paired": paired, "sam": {}}
        for fn in sorted(os.listdir(src)):
            if fn.endswith(".gz"):
                data = gzip.open(os.path.join(src, fn), "rb").read()
                tgt = dst / fn[:-3]
                tgt.write_bytes(data)
                if fn.startswith("r1."):
                    info["r1"] = str(tgt)
                elif fn.startswith("r2."):
                    info["r2"] = str(tgt)
                elif fn.startswith("ref.") and fn.endswith(".sam.gz"):
                    info["sam"][fn.split(".")[1]] = str(tgt)
                elif fn.startswith("ref.") and fn.endswith(".prof.gz"):
                    info["prof"] = (fn.split(".")[1], str(tgt), str(tgt)[:-5] + ".maps")
        info.setdefault("r2", None)
        out[name] = info
    return out


def sam_diff(path_a, path_b, limit=3):
    """Number of differing lines (+ a few examples)."""
    a = open(path_a, encoding="latin-1").read().split("\n")
    b = open(path_b, encoding="latin-1").read().split("\n")
    bad = [(x, y) for x, y in zip(a, b) if x != y]
    return len(bad) + abs(len(a) - len(b)), bad[:limit]
