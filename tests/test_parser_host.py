"""The product's sequential reader (mcx_files.cpp: Parser — ordinary .gz through the parallel inflater, bgzip's container, zlib's one thread, plain FASTA with
multi-line records) on the host, without a GPU, through tests/hostemu/parser_check.cpp: the records it hands out against a restatement of the reference's
text rules (GetData.cpp:3-20 header trimming, :45-55 / :101-128 record shapes, the 1024-byte line buffer of the .gz reader, the last byte of a sequence
line dropped) in Python — on files large enough that records straddle the feeder's 8 MB blocks at arbitrary offsets, with lines longer than the room a
block keeps in front, NUL bytes, a missing final newline."""
import ctypes
import gzip
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def parser_lib():
    d = os.path.join(ROOT, "tests", "hostemu")
    if not os.path.exists(os.path.join(ROOT, "mapcaller_amd", "libmcx.so")):
        pytest.skip("libmcx.so is not built")
    subprocess.run(["make", "-C", d, "libparser_check.so"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = ctypes.CDLL(os.path.join(d, "libparser_check.so"))
    L.parser_dump.restype = ctypes.c_longlong
    L.parser_dump.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    return L


def header(line):
    """IdentifyHeaderBegPos / IdentifyHeaderEndPos (GetData.cpp:3-20) on a line with its '\\n': from the first byte that is not '>' or '@' to the first blank,
    '/' or unprintable byte, looked for in the first 100 bytes."""
    n = len(line)
    p1 = next((i for i in range(1, n) if line[i] not in b">@"), n - 1)
    lim = min(n, 100)
    p2 = next((i for i in range(1, lim) if line[i] <= 0x20 or line[i] == 0x2F or line[i] >= 0x7F), lim - 1)
    return line[p1:p2] if p2 > p1 else b""


def expect_fastq(raw, gz):
    """name, bases, qualities of every record as the reference's readers cut them (a record is four lines, whatever they hold)."""
    lines = raw.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
        nl = [True] * len(lines)
    else:
        nl = [True] * (len(lines) - 1) + [False]
    out = []
    for i in range(0, len(lines) - 3, 4):
        h, s, q = lines[i], lines[i + 1], lines[i + 3]
        seq = s if nl[i + 1] else s[:-1]  # the last byte of the line (its '\n') is dropped: without one, a base goes
        qual = q[: len(seq)]
        out.append((header(h + b"\n"), seq, qual))
    return out


def dump(L, path, tmp_path, per_take=777, max_len=100000):
    out = str(tmp_path / "dump.txt")
    err = ctypes.create_string_buffer(512)
    n = L.parser_dump(path.encode(), max_len, per_take, out.encode(), err, 512)
    assert n >= 0, err.value
    recs = [tuple(l.split(b"\t")) for l in open(out, "rb").read().split(b"\n")[:-1]]
    assert len(recs) == n
    return recs


def make_fastq(rng, n, lo, hi):
    parts = []
    for i in range(n):
        m = rng.randint(lo, hi)
        seq = bytes(rng.choice(b"ACGTN") for _ in range(m)) if m < 400 else bytes(rng.choice(b"ACGT") for _ in range(64)) * (m // 64 + 1)
        seq = seq[:m]
        qual = bytes(33 + (j * 7 + i) % 40 for j in range(m))
        name = b"@r%d%s" % (i, rng.choice([b"", b" extra words", b"/1", b"\tx"]))
        parts.append(name + b"\n" + seq + b"\n+\n" + qual + b"\n")
    return b"".join(parts)


@pytest.mark.parametrize("kind", ["gz", "gz_zlib", "bgzf"])
def test_records_that_straddle_the_feeders_blocks(parser_lib, tmp_path, monkeypatch, kind):
    """30 MB of FASTQ text (reads of 30 to 300 bases, so that record boundaries fall everywhere relative to the 8 MB blocks), as an ordinary .gz, through zlib's
    one thread, and as bgzip's container: every record equals the restated rules'."""
    rng = random.Random(5)
    raw = make_fastq(rng, 125_000, 30, 300)
    assert len(raw) > 3 * (8 << 20)
    want = expect_fastq(raw, True)
    path = str(tmp_path / "r.fq.gz")
    if kind == "bgzf":
        import struct, zlib
        with open(path, "wb") as f:
            for i in range(0, len(raw), 0xff00):
                chunk = raw[i:i + 0xff00]
                c = zlib.compressobj(1, zlib.DEFLATED, -15)
                comp = c.compress(chunk) + c.flush()
                f.write(bytes.fromhex("1f8b08040000000000ff0600424302") + b"\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
            f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    else:
        with gzip.open(path, "wb", compresslevel=4) as f:
            f.write(raw)
    if kind == "gz_zlib":
        monkeypatch.setenv("MCX_GZ_SERIAL", "1")
    got = dump(parser_lib, path, tmp_path)
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, (k, a[0], b[0])


def test_lines_longer_than_a_blocks_room_in_front(parser_lib, tmp_path):
    """Plain multi-line FASTA (GetData.cpp:56-77) with records of up to 3 MB on lines of up to 200 KB — more than the 64 KB a block keeps in front for what the
    block before left unfinished — and a last line without a newline: the bases of every record, joined."""
    rng = random.Random(9)
    recs, parts = [], []
    for i in range(26):
        n_lines = rng.randint(1, 30)
        lines = [bytes(rng.choice(b"ACGT") for _ in range(50)) * (rng.randint(1, 4000)) for _ in range(n_lines)]
        lines = [l[: rng.randint(1, len(l))] for l in lines]
        recs.append((b"s%d" % i, b"".join(lines)))
        parts.append(b">s%d some words\n" % i + b"\n".join(lines) + b"\n")
    raw = b"".join(parts)[:-1]  # no final newline: the reference drops a line's last byte all the same
    recs[-1] = (recs[-1][0], recs[-1][1][:-1])
    assert len(raw) > 2 * (8 << 20)
    path = str(tmp_path / "g.fa")
    open(path, "wb").write(raw)
    got = dump(parser_lib, path, tmp_path, per_take=3, max_len=1 << 30)
    assert [(a[0], a[1]) for a in got] == recs


def test_a_nul_cuts_a_gz_line_short_in_whatever_block_it_lies(parser_lib, tmp_path):
    """gzgets' lines are C strings (GetData.cpp:101-128): a NUL inside a sequence line of a .gz input ends the line there — strlen semantics, and the byte
    before it goes as the '\\n' would (:113).  One in the first block, one 20 MB on; every other record is untouched."""
    rng = random.Random(11)
    raw = bytearray(make_fastq(rng, 120_000, 100, 250))
    lines = bytes(raw).split(b"\n")
    want = expect_fastq(bytes(raw), True)
    cut_at = {}
    for rec in (10, 100_000):
        s = lines[4 * rec + 1]
        cut = len(s) // 2
        raw[sum(len(l) + 1 for l in lines[: 4 * rec + 1]) + cut] = 0
        cut_at[rec] = cut
    path = str(tmp_path / "n.fq.gz")
    with gzip.open(path, "wb", compresslevel=4) as f:
        f.write(bytes(raw))
    got = dump(parser_lib, path, tmp_path)
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        if k in cut_at:
            cut = cut_at[k]
            assert a == (b[0], lines[4 * k + 1][: cut - 1], lines[4 * k + 3][: cut - 1]), (k, a)
        else:
            assert a == b, (k, a[0], b[0])
