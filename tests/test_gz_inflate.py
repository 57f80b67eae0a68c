"""The file front end's parallel reader for ordinary .gz input (mapcaller_amd/csrc/mcx_pgz.h) against zlib: the same bytes, whatever the number of threads and
the length of the stretches the stream is cut into.  (Replaces the reference's gzgets loop, src/GetData.cpp:101-146, for plain gzip streams; CPU only.)"""
import ctypes as C
import gzip
import os
import random
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def inflate():
    from mapcaller_amd import api
    L = api.lib()
    L.mcx_gz_inflate.restype = C.c_int64
    L.mcx_gz_inflate.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]

    def run(path, threads, stretch, room):
        buf = np.zeros(room + 16, dtype=np.uint8)
        n = C.c_uint64()
        r = L.mcx_gz_inflate(str(path).encode(), threads, stretch, buf.ctypes.data, buf.size, C.byref(n))
        return r, bytes(buf[:max(0, min(r, room))]) if r >= 0 else b"", n.value
    return run


def fastq(n, seed, rlen=150):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        seq = "".join(rng.choice("ACGT") for _ in range(rlen))
        q = "".join("F" if rng.random() < 0.9 else rng.choice(":,#") for _ in range(rlen))
        out.append(f"@A00123:45:HXXXXXXXX:1:{1101 + i // 1000}:{1000 + i % 977}:{i % 7} {1 + i % 2}:N:0:ACGT\n{seq}\n+\n{q}\n")
    return "".join(out).encode()


@pytest.mark.parametrize("name", ["toy", "var", "mc", "long"])
def test_golden_read_files(inflate, name):
    p = os.path.join(GOLD, name, "r1.fq.gz")
    want = gzip.open(p, "rb").read()
    for threads, stretch in ((1, 0), (3, 1 << 16), (8, 1 << 16)):
        r, got, _ = inflate(p, threads, stretch, len(want))
        assert r == len(want) and got == want, (name, threads, stretch, r)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_fastq_at_every_level_threads_and_stretch(inflate, tmp_path, level):
    data = fastq(40_000, 7 + level)
    p = tmp_path / f"l{level}.fq.gz"
    with gzip.open(p, "wb", compresslevel=level) as f:
        f.write(data)
    for threads, stretch in ((1, 0), (2, 1 << 20), (4, 1 << 18), (8, 1 << 16), (5, 100_000)):
        r, got, _ = inflate(p, threads, stretch, len(data))
        assert r == len(data) and got == data, (level, threads, stretch, r)


def test_members_stored_and_fixed_blocks_and_odd_streams(inflate, tmp_path):
    data = fastq(20_000, 3)
    # several members behind one another (RFC 1952 2.2), one of them empty
    p = tmp_path / "mm.fq.gz"
    with open(p, "wb") as f:
        for k in range(4):
            f.write(gzip.compress(data[k * 1_000_000:(k + 1) * 1_000_000], 6))
            if k == 1:
                f.write(gzip.compress(b""))
    r, got, _ = inflate(p, 4, 1 << 16, 4_000_000)
    assert r == 4_000_000 and got == data[:4_000_000]
    # stored blocks (level 0), fixed-Huffman blocks (tiny inputs), a header with a file name and a comment
    for tag, blob in (("stored", gzip.compress(data[:300_000], 0)), ("tiny", gzip.compress(b"@r\nACGT\n+\nFFFF\n", 9))):
        q = tmp_path / f"{tag}.gz"
        q.write_bytes(blob)
        want = gzip.decompress(blob)
        r, got, _ = inflate(q, 4, 1 << 16, len(want))
        assert r == len(want) and got == want, tag
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(data[:500_000]) + co.flush()
    hdr = b"\x1f\x8b\x08\x18" + b"\0\0\0\0" + b"\x00\x03" + b"reads.fq\0" + b"a comment\0"
    q = tmp_path / "named.gz"
    q.write_bytes(hdr + body + (zlib.crc32(data[:500_000]) & 0xFFFFFFFF).to_bytes(4, "little") + (500_000).to_bytes(4, "little"))
    r, got, _ = inflate(q, 4, 1 << 16, 500_000)
    assert r == 500_000 and got == data[:500_000]
    # bytes that are not text: no block start is ever accepted, one thread inflates a round — still the right bytes
    blob = os.urandom(1 << 18) * 3 + bytes(range(256)) * 2000
    q = tmp_path / "bin.gz"
    q.write_bytes(gzip.compress(blob, 6))
    r, got, _ = inflate(q, 4, 1 << 16, len(blob))
    assert r == len(blob) and got == blob


def test_a_damaged_stream_is_refused(inflate, tmp_path):
    data = fastq(30_000, 11)
    raw = bytearray(gzip.compress(data, 6))
    raw[len(raw) // 2] ^= 0x55
    p = tmp_path / "bad.fq.gz"
    p.write_bytes(bytes(raw))
    r, _, delivered = inflate(p, 4, 1 << 16, len(data))
    assert r == -2 and delivered < len(data)
    # a wrong CRC in the trailer: everything inflates, the member is refused all the same
    raw = bytearray(gzip.compress(data, 6))
    raw[-6] ^= 1
    p.write_bytes(bytes(raw))
    r, _, _ = inflate(p, 4, 1 << 16, len(data))
    assert r == -2
    # cut short
    raw = gzip.compress(data, 6)
    p.write_bytes(raw[: len(raw) * 2 // 3])
    r, _, _ = inflate(p, 4, 1 << 16, len(data))
    assert r == -2
    # not a gzip file at all
    p.write_bytes(data[:1000])
    r, _, _ = inflate(p, 4, 1 << 16, 1000)
    assert r == -1
