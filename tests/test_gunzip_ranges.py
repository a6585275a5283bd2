"""RangeGunzip (nohuman_amd/csrc/nh_inflate.cpp, round 6): the host's share of the HYBRID gzip reader -- while the GPU's codec
kernels are the run's bottleneck, some cells of the stream's piece grid are inflated by the host's cores (nh_gunzip.hip).  A cell
is decoded speculatively, every chunk at once, BEFORE the stream's position and window at the cell are known, and stitched when
they are.  Checked here without a GPU through nh_debug_gunzip_ranges: the file as a chain of cells, each by a fresh RangeGunzip
(or every second / third one, the others by the sequential decoder, standing in for the GPU's pieces): whatever the cell and
chunk sizes and the stream's shape, the bytes are zlib's and the members' CRCs come out right; damage is reported."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from nohuman_amd import _lib
from tests.test_codec import fastq_like
from tests.test_gunzip import deflate_raw

FASTQ = fastq_like(5_000_000, seed=33)


def ranges(src, dst, threads, cell, chunk, every=1):
    L = _lib.lib()
    fn = L.nh_debug_gunzip_ranges
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    st = (C.c_uint64 * 4)()
    rc = fn(os.fsencode(src), os.fsencode(dst), threads, cell, chunk, every, st)
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())
    return list(st)


def check(tmp_path, raw_gz, want, threads, cell, chunk, every=1, name="x"):
    src, dst = tmp_path / (name + ".gz"), tmp_path / (name + ".out")
    src.write_bytes(raw_gz)
    st = ranges(src, dst, threads, cell, chunk, every)
    got = dst.read_bytes()
    assert len(got) == len(want) and got == want, (threads, cell, chunk, every)
    return st


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("threads,cell,chunk,every", [(1, 1 << 20, 1 << 18, 1), (4, 300_000, 50_000, 1), (3, 200_000, 70_000, 2),
                                                       (8, 1 << 19, 1 << 16, 3), (2, 64_000, 3_000, 2), (4, 1 << 24, 100_000, 1)])
def test_fastq_by_cells(tmp_path, level, threads, cell, chunk, every):
    st = check(tmp_path, gzip.compress(FASTQ, level), FASTQ, threads, cell, chunk, every)
    assert st[0] >= 1
    if cell == 300_000 and level == 6:
        assert st[0] > 3 and st[1] > 3 * st[0]  # several cells, and their chunks really were decoded ahead and accepted


@pytest.mark.parametrize("kind", ["zeros", "period3", "random", "text", "tiny", "empty"])
def test_shapes_of_data(tmp_path, kind):
    rng = np.random.default_rng(5)
    data = {
        "zeros": bytes(3_000_000),
        "period3": b"ACG" * 700_000,
        "random": rng.integers(0, 256, 1_500_000, dtype=np.uint8).tobytes(),  # stored blocks: nothing for the search to find
        "text": b"".join(b"line %d of some text\n" % i for i in range(120_000)),
        "tiny": b"hello, world\n",
        "empty": b"",
    }[kind]
    for threads, cell, chunk, every in ((1, 100_000, 20_000, 1), (4, 50_000, 2_000, 2), (3, 1 << 20, 50_000, 1)):
        check(tmp_path, gzip.compress(data, 6), data, threads, cell, chunk, every, kind)


@pytest.mark.parametrize("strategy", [zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_FILTERED])
def test_zlib_strategies(tmp_path, strategy):
    data = FASTQ[:1_500_000]
    check(tmp_path, deflate_raw(data, 6, strategy), data, 3, 150_000, 40_000, 2)


@pytest.mark.parametrize("flush", [zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])
def test_flush_points(tmp_path, flush):
    data = FASTQ[:2_000_000]
    check(tmp_path, deflate_raw(data, 6, flush_every=10_000, flush=flush), data, 4, 120_000, 30_000, 2)


def test_many_members_and_trailing_bytes(tmp_path):
    """members end inside cells, at their edges and by the dozen in one chunk: the stretches the ranges report carry every member's
    books; bytes behind the last member are ignored like gzip ignores them"""
    parts, want = [], []
    for i in range(0, 3_000_000, 60_000):
        blk = FASTQ[i:i + 60_000]
        parts.append(gzip.compress(blk, 6))
        want.append(blk)
    for i in range(40):
        blk = b"tiny member %d\n" % i
        parts.append(gzip.compress(blk))
        want.append(blk)
    parts.append(gzip.compress(b""))
    raw, text = b"".join(parts), b"".join(want)
    for threads, cell, chunk, every in ((4, 100_000, 30_000, 1), (3, 250_000, 8_000, 2), (2, 1 << 20, 1 << 18, 1)):
        check(tmp_path, raw, text, threads, cell, chunk, every)
        check(tmp_path, raw + b"\0\0trailing garbage", text, threads, cell, chunk, every)


def test_damage_is_reported(tmp_path):
    good = gzip.compress(FASTQ[:2_000_000], 6)
    for where in (len(good) // 3, len(good) // 2, len(good) - 6):  # data, data, the stored crc
        bad = bytearray(good)
        bad[where] ^= 0x41
        p = tmp_path / "bad.gz"
        p.write_bytes(bytes(bad))
        with pytest.raises(RuntimeError):
            ranges(p, tmp_path / "bad.out", 4, 200_000, 50_000, 2)
    p = tmp_path / "cut.gz"
    p.write_bytes(good[: len(good) * 2 // 3])
    with pytest.raises(RuntimeError):
        ranges(p, tmp_path / "cut.out", 4, 200_000, 50_000, 1)


def test_random_streams_against_zlib(tmp_path):
    rng = np.random.default_rng(77)
    for case in range(12):
        segs = []
        for _ in range(int(rng.integers(2, 9))):
            kind = int(rng.integers(0, 5))
            n = int(rng.integers(1_000, 400_000))
            if kind == 0:
                segs.append(FASTQ[int(rng.integers(0, 1_000_000)):][:n])
            elif kind == 1:
                segs.append(rng.integers(0, 256, n // 4, dtype=np.uint8).tobytes())
            elif kind == 2:
                segs.append(bytes([int(rng.integers(65, 70))]) * n)
            elif kind == 3:
                segs.append((b"ACGT" * 50 + b"\n") * (n // 201 + 1))
            else:
                segs.append(b"")
        data = b"".join(segs)
        level = int(rng.choice([1, 4, 6, 9]))
        raw = gzip.compress(data, level) if rng.random() < 0.5 else deflate_raw(data, level, flush_every=int(rng.integers(5_000, 90_000)))
        cell = int(rng.integers(20_000, 600_000))
        check(tmp_path, raw, data, int(rng.integers(1, 7)), cell, int(rng.integers(2_000, cell + 1)), int(rng.integers(1, 4)), "r%d" % case)
