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
SETS = {"toy": True, "mc": True, "se": False, "long": True, "var": True}  # name -> paired
# the reference's -vcf runs kept as ref.vcf.<tag>.gz (scripts/make_golden.py): tag -> (alg, extra command-line options)
VCF_RUNS = {
    "default": ("ksw2", []), "nw": ("nw", []), "gvcf": ("ksw2", ["-gvcf"]), "mono": ("ksw2", ["-monomorphic"]),
    "filter": ("ksw2", ["-filter"]), "ploidy1": ("ksw2", ["-ploidy", "1"]), "somatic": ("ksw2", ["-somatic"]),
    "opts": ("ksw2", ["-ad", "3", "-min_gap", "20", "-min_cnv", "20", "-size", "400", "-dup", "3", "-maxclip", "10", "-id", "s1"]),
}
VCF_CASES = [("toy", "default"), ("mc", "default"), ("se", "default"), ("long", "default")] + [("var", t) for t in VCF_RUNS]


def vcf_alg(name, tag):
    """-alg of a golden VCF run (the mc set's default run used nw)."""
    return "nw" if (name, tag) == ("mc", "default") else VCF_RUNS[tag][0]


def vcf_body(path):
    """VCF text without the two header lines that hold paths."""
    return [l for l in open(path, encoding="latin-1").read().split("\n") if not l.startswith(("##command_line=", "##reference="))]


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
    L.mcxo_pair_totals.restype = ctypes.c_int64
    L.mcxo_pair_totals.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    L.mcxo_map_files_interleaved.restype = ctypes.c_int64
    L.mcxo_map_files_interleaved.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    L.mcxo_map_files_profile.restype = ctypes.c_int64
    L.mcxo_map_files_profile.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.mcxo_vcf_defaults.argtypes = [ctypes.c_void_p]
    L.mcxo_map_files_vcf.restype = ctypes.c_int64
    L.mcxo_map_files_vcf.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p]
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
def hostemu_variants_lib():
    import ctypes
    d = os.path.join(ROOT, "tests", "hostemu")
    _make(d, "libhostemu_variants.so")
    L = ctypes.CDLL(os.path.join(d, "libhostemu_variants.so"))
    L.hostemu_call_variants.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p, ctypes.c_char_p]
    L.hostemu_vc_error.restype = ctypes.c_char_p
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
        info = {"prefix": str(dst / "idx"), "paired": paired, "sam": {}, "vcf": {}}
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
                elif fn.startswith("ref.vcf."):
                    info["vcf"][fn.split(".")[2]] = str(tgt)
                elif fn.startswith("ref.") and fn.endswith(".prof.gz"):
                    info["prof"] = (fn.split(".")[1], str(tgt), str(tgt)[:-5] + ".maps")
        info.setdefault("r2", None)
        out[name] = info
    return out


@pytest.fixture(scope="session")
def io_golden(tmp_path_factory, golden):
    """tests/golden/io unpacked: input-side cases on the toy index (interleaved -p, multi-line FASTA,
    .gz handed over directly, two libraries)."""
    src = os.path.join(GOLD, "io")
    dst = tmp_path_factory.mktemp("io")
    out = {"prefix": golden["toy"]["prefix"], "gz1": os.path.join(GOLD, "toy", "r1.fq.gz"), "gz2": os.path.join(GOLD, "toy", "r2.fq.gz")}
    for fn in sorted(os.listdir(src)):
        tgt = dst / fn[:-3]
        tgt.write_bytes(gzip.open(os.path.join(src, fn), "rb").read())
        out[fn[:-3]] = str(tgt)
    # the two libraries: first 700 pairs, the rest
    for tag in ("1", "2"):
        lines = open(golden["toy"]["r" + tag], "rb").read().split(b"\n")
        (dst / f"a{tag}.fq").write_bytes(b"\n".join(lines[:2800]) + b"\n")
        (dst / f"b{tag}.fq").write_bytes(b"\n".join(lines[2800:6000]) + b"\n")
        out["a" + tag], out["b" + tag] = str(dst / f"a{tag}.fq"), str(dst / f"b{tag}.fq")
    return out


def maps_canon(text):
    """The .maps text with the inversion / translocation site lists ordered by (position, distance):
    the reference orders them with std::sort on the position alone (ReadMapping.cpp:627-628), so
    the order among equal positions is not defined."""
    head, sites = [], []
    for l in text.split("\n"):
        if l[:2] in ("V ", "T "):
            f = l.split()
            sites.append((f[0], int(f[1]), int(f[2])))
        elif l:
            head.append(l)
    sites.sort(key=lambda t: (t[0] != "V", t[1], t[2]))
    return head + [f"{a} {b} {c}" for a, b, c in sites]


class VcfOpts:
    """The option block both libraries take (oracle mcxo_vcf_opts == product mcx_vcf_opts), filled from
    MapCaller's command-line switches."""

    def __init__(self, args):
        import ctypes

        class S(ctypes.Structure):
            _fields_ = [(k, ctypes.c_int) for k in ("ploidy", "min_allele_depth", "min_cnv", "min_gap", "fragment_size", "filter", "gvcf",
                                                   "monomorphic", "somatic", "max_dup", "max_clip")] + \
                       [("freq_thr", ctypes.c_float), ("sample_id", ctypes.c_char_p), ("ref_name", ctypes.c_char_p), ("cmdline", ctypes.c_char_p)]
        o = S(ploidy=2, min_allele_depth=5, min_cnv=50, min_gap=50, fragment_size=500, max_dup=5, max_clip=5, freq_thr=0.2,
              sample_id=b"unknown", ref_name=b"ref", cmdline=b"test")
        it = iter(args)
        for a in it:
            if a == "-gvcf": o.gvcf = 1
            elif a == "-monomorphic": o.monomorphic = 1
            elif a == "-filter": o.filter = 1
            elif a == "-somatic": o.somatic = 1
            elif a == "-ploidy": o.ploidy = min(2, int(next(it)))
            elif a == "-ad": o.min_allele_depth = int(next(it))
            elif a == "-min_cnv": o.min_cnv = int(next(it))
            elif a == "-min_gap": o.min_gap = int(next(it))
            elif a == "-size": o.fragment_size = int(next(it))
            elif a == "-dup": o.max_dup = int(next(it))
            elif a == "-maxclip": o.max_clip = int(next(it))
            elif a == "-id": o.sample_id = next(it).encode()
            else: raise ValueError(a)
        self.struct = o
        self.ref = ctypes.byref(o)


def _mask_se_reverse_qual(line):
    f = line.split("\t")
    if len(f) > 10 and f[1].isdigit() and (int(f[1]) & 0x11) == 0x10:
        f[10] = "?"
    return "\t".join(f)


def sam_diff(path_a, path_b, limit=3, mask_se_reverse_qual=False):
    """Number of differing lines (+ a few examples).  mask_se_reverse_qual: the reference prints an
    uninitialised first quality byte for reverse-strand single-end FASTQ reads (SamReport.cpp:318-322),
    so QUAL of those lines is not compared."""
    a = open(path_a, "rb").read().decode("latin-1").split("\n")  # (binary: no newline translation of stray bytes)
    b = open(path_b, "rb").read().decode("latin-1").split("\n")
    if mask_se_reverse_qual:
        a = [_mask_se_reverse_qual(l) for l in a]
        b = [_mask_se_reverse_qual(l) for l in b]
    bad = [(x, y) for x, y in zip(a, b) if x != y]
    return len(bad) + abs(len(a) - len(b)), bad[:limit]
