"""The gzip READER on the GPU (nh_gunzip_device_file / nohuman_amd/csrc/nh_gunzip.hip): every stream of
tests/test_gunzip.py -- levels, strategies, flush points, stored / fixed blocks, many members, header fields,
trailing bytes, damage -- through block search, decode, window scan, marker resolve and CRC-32 on the device.
Whatever the piece and chunk sizes, the bytes must equal what zlib produces, and damaged files must be reported,
not passed on."""
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

pytestmark = pytest.mark.gpu


def gunzip_dev(src, dst, seg=0, stretch=0):
    L = _lib.lib()
    st = (C.c_uint64 * 8)()
    rc = L.nh_gunzip_device_file(os.fsencode(src), os.fsencode(dst), 0, seg, stretch, st)
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())
    return dict(zip(("pieces", "chunks", "redecoded", "host_pieces", "members", "text", "gzip", "kernel_us"), st))


def check(tmp_path, raw_gz, want, seg=0, stretch=0, name="x"):
    src, dst = tmp_path / (name + ".gz"), tmp_path / (name + ".out")
    src.write_bytes(raw_gz)
    st = gunzip_dev(src, dst, seg, stretch)
    got = dst.read_bytes()
    assert len(got) == len(want), (len(got), len(want), st)
    if got != want:
        a = np.frombuffer(got, np.uint8)
        b = np.frombuffer(want, np.uint8)
        bad = np.nonzero(a != b)[0]
        raise AssertionError("%d bytes differ, first at %d (%s)" % (bad.size, bad[0], st))
    assert st["text"] == len(want)
    return st


FASTQ = fastq_like(6_000_000, seed=21)
SHAPES = [(0, 0), (1 << 20, 32768), (300_000, 8192), (64_000, 4096), (40_000, 2048)]


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("seg,stretch", SHAPES)
def test_fastq_levels_pieces_chunks(tmp_path, level, seg, stretch):
    st = check(tmp_path, gzip.compress(FASTQ, level), FASTQ, seg, stretch)
    assert st["host_pieces"] == 0 and st["members"] == 1
    if stretch == 8192:
        assert st["chunks"] > 20 and st["pieces"] > 3  # chunks really were decoded apart and chained by the scan


@pytest.mark.parametrize("kind", ["zeros", "run_a", "period3", "random", "text", "tiny", "empty", "one"])
def test_shapes_of_data(tmp_path, kind):
    rng = np.random.default_rng(5)
    data = {
        "zeros": bytes(3_000_000),
        "run_a": b"A" * 2_500_001,
        "period3": b"ACG" * 700_000,
        "random": rng.integers(0, 256, 1_500_000, dtype=np.uint8).tobytes(),  # stored blocks
        "text": b"".join(b"line %d of some text\n" % i for i in range(120_000)),
        "tiny": b"hello, world\n",  # fixed Huffman block
        "empty": b"",
        "one": b"x",
    }[kind]
    for seg, stretch in ((0, 0), (30_000, 2048), (200_000, 16384)):
        st = check(tmp_path, gzip.compress(data, 6), data, seg, stretch, kind)
        if kind in ("zeros", "run_a", "period3"):
            assert st["host_pieces"] > 0  # text beyond 16 : 1 per chunk: the host decoder takes over, loudly (stderr)


@pytest.mark.parametrize("strategy", [zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_FILTERED])
def test_zlib_strategies(tmp_path, strategy):
    data = FASTQ[:1_500_000]
    check(tmp_path, deflate_raw(data, 6, strategy), data, 100_000, 4096)
    check(tmp_path, deflate_raw(data, 6, strategy), data)


@pytest.mark.parametrize("flush", [zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])
def test_flush_points_and_empty_stored_blocks(tmp_path, flush):
    data = FASTQ[:2_000_000]
    check(tmp_path, deflate_raw(data, 6, flush_every=10_000, flush=flush), data, 90_000, 4096)


def test_many_small_members_and_header_fields(tmp_path):
    """Members of 60 kB with extra fields (no BGZF: no 'BC' subfield; the real thing is test_bgzf_chunk_starts_... below),
    a member with name, comment and header CRC, an empty member at the end.  More member ends than a chunk records
    (four) send a piece to the host decoder; the bytes and the CRC checks are the same."""
    parts, want = [], []
    for i in range(0, 3_000_000, 60_000):
        blk = FASTQ[i:i + 60_000]
        body = zlib.compress(blk, 6)[2:-4]
        extra = b"XY\x02\x00" + struct.pack("<H", 0)
        hdr = b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", len(extra)) + extra
        parts.append(hdr + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        want.append(blk)
    blk = b"tail member\n" * 1000
    hdr = b"\x1f\x8b\x08\x1a" + bytes(4) + b"\x00\x03" + b"name.fq\x00" + b"a comment\x00" + b"\x12\x34"
    parts.append(hdr + zlib.compress(blk, 9)[2:-4] + struct.pack("<II", zlib.crc32(blk), len(blk)))
    want.append(blk)
    parts.append(gzip.compress(b""))
    raw, data = b"".join(parts), b"".join(want)
    assert gzip.decompress(raw) == data
    for seg, stretch in ((0, 0), (200_000, 8192), (50_000, 2048)):
        st = check(tmp_path, raw, data, seg, stretch)
        assert st["members"] == 52


def bgzf(data, block=65280, level=6, eof=True):
    """bgzip's format (htslib): members of at most 64 KiB of text, each ONE final deflate block, its whole size - 1 in the 'B' 'C'
    subfield of the header's extra field; an empty member marks the end"""
    out = []
    for i in list(range(0, len(data), block)) + ([None] if eof else []):
        blk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(blk) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body +
                   struct.pack("<II", zlib.crc32(blk), len(blk)))
    return b"".join(out)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_bgzf_chunk_starts_come_from_the_member_headers(tmp_path, level):
    """BGZF on the device: no block search -- every member is one FINAL block, which the search does not
    look for -- the chunks' starts are read off the 'B' 'C' sizes, and no piece goes to the host decoder."""
    raw = bgzf(FASTQ, level=level)
    assert gzip.decompress(raw) == FASTQ
    for seg, stretch in ((0, 0), (1 << 20, 16384), (200_000, 8192), (64_000, 32768)):
        st = check(tmp_path, raw, FASTQ, seg, stretch, "bgzf")
        assert st["host_pieces"] == 0 and st["members"] == len(FASTQ) // 65280 + 2 and st["redecoded"] == 0
    # small members (many a chunk: the host decoder takes those pieces over, loudly), no end marker, damage
    small = bgzf(FASTQ[:1_500_000], block=3000, eof=False)
    check(tmp_path, small, FASTQ[:1_500_000], 0, 0, "bgzf_small")
    bad = bytearray(raw)
    bad[len(raw) // 2] ^= 0x10
    src = tmp_path / "bgzf_bad.gz"
    src.write_bytes(bytes(bad))
    with pytest.raises(RuntimeError):
        gunzip_dev(src, tmp_path / "bgzf_bad.out")
    # a BGZF head on an ordinary gzip tail (cat of two tools' outputs): the headers stop saying sizes, the search takes over
    mixed = bgzf(FASTQ[:1_000_000], eof=False) + gzip.compress(FASTQ[1_000_000:], 6)
    check(tmp_path, mixed, FASTQ, 300_000, 8192, "bgzf_mixed")


def test_members_of_whole_files_concatenated(tmp_path):
    """cat a.gz b.gz c.gz: members of different levels, the chunks chain across the member ends (a member's start
    inside a chunk cuts the window chain) and every member's CRC-32 and length are checked."""
    a, b, c = FASTQ[:2_000_000], FASTQ[2_000_000:2_700_000], FASTQ[2_700_000:]
    raw = gzip.compress(a, 6) + gzip.compress(b, 1) + gzip.compress(c, 9)
    for seg, stretch in ((0, 0), (150_000, 4096)):
        st = check(tmp_path, raw, FASTQ, seg, stretch)
        assert st["members"] == 3 and st["host_pieces"] == 0


def test_trailing_bytes_after_the_last_member_are_ignored(tmp_path):
    data = FASTQ[:400_000]
    check(tmp_path, gzip.compress(data) + bytes(1000), data, 50_000, 2048)
    check(tmp_path, gzip.compress(data) + b"this is not gzip", data, 50_000, 2048)
    check(tmp_path, gzip.compress(data) + b"this is not gzip", data)


def test_long_distance_and_long_match_edges(tmp_path):
    rng = np.random.default_rng(8)
    block = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    data = block + block + block[:258] * 50 + block[::-1] + block  # distance 32768, maximal matches
    check(tmp_path, gzip.compress(data, 9), data, 20_000, 2048)
    check(tmp_path, gzip.compress(data, 9), data)


def test_a_false_block_start_is_decoded_again_from_the_true_end(tmp_path, monkeypatch):
    """NOHUMAN_GZDEV_FAKE_START=c plants a start in stretch c that is no block boundary: the chunk before it does not end
    there, the seam check strikes it and the predecessor is decoded again to the next start (broken seam)."""
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_START", "5")
    st = check(tmp_path, gzip.compress(FASTQ, 6), FASTQ, 1 << 20, 16384)
    assert st["redecoded"] >= 1 and st["host_pieces"] == 0


def test_a_false_start_whose_garbage_ends_the_stream_loses_no_text(tmp_path, monkeypatch):
    """The planted false start also claims the end of the stream (NOHUMAN_GZDEV_FAKE_END): the chunks behind it are struck
    for a moment, the seam check strikes the false start itself, and the chunks behind must come back WITH what they
    decoded (a 7.4 GB bench input lost 115 MB of one piece this way: crc error; tools/gz_debug.py found it)."""
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_START", "5")
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_END", "1")
    st = check(tmp_path, gzip.compress(FASTQ, 6), FASTQ, 1 << 20, 16384)
    assert st["redecoded"] >= 1 and st["host_pieces"] == 0


def test_damage_is_reported(tmp_path):
    good = gzip.compress(FASTQ[:3_000_000], 6)
    src, dst = tmp_path / "bad.gz", tmp_path / "bad.out"
    for seg, stretch in ((0, 0), (100_000, 4096)):
        for where in (len(good) // 3, len(good) // 2, len(good) - 6):  # data, data, stored crc
            bad = bytearray(good)
            bad[where] ^= 0x55
            src.write_bytes(bytes(bad))
            with pytest.raises(RuntimeError):
                gunzip_dev(src, dst, seg, stretch)
        src.write_bytes(good[: len(good) // 2])  # truncated
        with pytest.raises(RuntimeError):
            gunzip_dev(src, dst, seg, stretch)
        bad = bytearray(good)
        bad[-2] ^= 0x01  # ISIZE
        src.write_bytes(bytes(bad))
        with pytest.raises(RuntimeError):
            gunzip_dev(src, dst, seg, stretch)
    src.write_bytes(b"\x1f\x8b\x08")  # header cut short
    with pytest.raises(RuntimeError):
        gunzip_dev(src, dst)
    src.write_bytes(b"@r1\nACGT\n+\nIIII\n" * 10)  # not gzip at all
    with pytest.raises(RuntimeError):
        gunzip_dev(src, dst)


def test_a_hundred_megabytes_of_fastq_in_default_pieces(tmp_path):
    """several default pieces (64 MiB of gzip each would be one: 8 MiB pieces here), two thousand chunks"""
    text = b"".join(fastq_like(12_000_000, seed=s) for s in range(9))
    st = check(tmp_path, gzip.compress(text, 6), text, 8 << 20, 0)
    assert st["pieces"] >= 2 and st["host_pieces"] == 0 and st["redecoded"] == 0
