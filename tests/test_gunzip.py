"""Multi-threaded gzip input decoder (nh_gunzip_file / nohuman_amd/csrc/nh_inflate.cpp) without a GPU.

kraken2 reads .gz inputs through `gzip -dc`; the host pipeline here decodes them in
process on several cores.  Whatever the chunking, the worker count or the shape of the deflate
stream, the bytes must equal what zlib produces, and damaged files must be reported, not passed on."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from nohuman_amd import _lib
from tests.test_codec import GZIP, compress, fastq_like


def gunzip(src, dst, threads, chunk):
    L = _lib.lib()
    st = (C.c_uint64 * 3)()
    rc = L.nh_gunzip_file(os.fsencode(src), os.fsencode(dst), threads, chunk, st)
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())
    return list(st)


def check(tmp_path, raw_gz, want, threads, chunk, name="x"):
    src, dst = tmp_path / (name + ".gz"), tmp_path / (name + ".out")
    src.write_bytes(raw_gz)
    st = gunzip(src, dst, threads, chunk)
    got = dst.read_bytes()
    assert len(got) == len(want)
    assert got == want
    return st


def deflate_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None, flush=zlib.Z_SYNC_FLUSH):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = []
    for i in range(0, len(data), flush_every):
        out.append(c.compress(data[i:i + flush_every]))
        out.append(c.flush(flush))
    out.append(c.flush())
    return b"".join(out)


FASTQ = fastq_like(6_000_000, seed=21)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("threads,chunk", [(1, 0), (2, 300_000), (3, 1 << 20), (4, 70_000), (3, 3_000), (8, 200_000)])
def test_fastq_levels_threads_chunks(tmp_path, level, threads, chunk):
    st = check(tmp_path, gzip.compress(FASTQ, level), FASTQ, threads, chunk)
    if chunk == 70_000 and threads == 4:
        assert st[0] > 10  # chunks really were decoded out of order and stitched


def test_markers_survive_whole_chunks_and_resolve(tmp_path):
    """Every record repeats the header text of the one before: back-references chain through the
    unknown window for the whole chunk, so the 16-bit path and its marker replacement carry it."""
    st = check(tmp_path, gzip.compress(FASTQ, 6), FASTQ, 4, 150_000)
    assert st[0] >= 8 and st[1] == 0


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
    for threads, chunk in ((1, 0), (3, 2_000), (4, 50_000)):
        check(tmp_path, gzip.compress(data, 6), data, threads, chunk, kind)


@pytest.mark.parametrize("strategy", [zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_FILTERED])
def test_zlib_strategies(tmp_path, strategy):
    data = FASTQ[:1_500_000]
    check(tmp_path, deflate_raw(data, 6, strategy), data, 3, 40_000)


@pytest.mark.parametrize("flush", [zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])
def test_flush_points_and_empty_stored_blocks(tmp_path, flush):
    data = FASTQ[:2_000_000]
    check(tmp_path, deflate_raw(data, 6, flush_every=10_000, flush=flush), data, 4, 30_000)


def test_output_of_our_parallel_compressor(tmp_path):
    """Blocks primed with a dictionary, separated by sync flushes, one member (nh_compress_file)."""
    src = tmp_path / "in.fq"
    src.write_bytes(FASTQ)
    gz = tmp_path / "mid.gz"
    compress(src, gz, GZIP, 4)
    check(tmp_path, gz.read_bytes(), FASTQ, 4, 100_000)


def test_many_members_bgzf_like_and_header_fields(tmp_path):
    parts, want = [], []
    for i in range(0, 3_000_000, 60_000):  # BGZF-sized members, each with an extra field
        blk = FASTQ[i:i + 60_000]
        body = zlib.compress(blk, 6)[2:-4]
        extra = b"BC\x02\x00" + struct.pack("<H", 0)
        hdr = b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", len(extra)) + extra
        parts.append(hdr + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        want.append(blk)
    # a member with file name, comment and header crc, then an empty member (bgzip's end marker)
    blk = b"tail member\n" * 1000
    hdr = b"\x1f\x8b\x08\x1a" + bytes(4) + b"\x00\x03" + b"name.fq\x00" + b"a comment\x00" + b"\x12\x34"
    parts.append(hdr + zlib.compress(blk, 9)[2:-4] + struct.pack("<II", zlib.crc32(blk), len(blk)))
    want.append(blk)
    parts.append(gzip.compress(b""))
    raw, data = b"".join(parts), b"".join(want)
    assert gzip.decompress(raw) == data
    for threads, chunk in ((1, 0), (4, 100_000), (3, 7_000)):
        check(tmp_path, raw, data, threads, chunk)


def test_trailing_bytes_after_the_last_member_are_ignored(tmp_path):
    data = FASTQ[:400_000]
    raw = gzip.compress(data) + bytes(1000)  # zero padding, as tape / some pipelines leave it
    check(tmp_path, raw, data, 3, 20_000)
    raw = gzip.compress(data) + b"this is not gzip"
    check(tmp_path, raw, data, 3, 20_000)


def test_long_distance_and_long_match_edges(tmp_path):
    rng = np.random.default_rng(8)
    block = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    data = block + block + block[:258] * 50 + block[::-1] + block  # distance 32768, maximal matches
    check(tmp_path, gzip.compress(data, 9), data, 2, 5_000)
    check(tmp_path, gzip.compress(data, 9), data, 4, 5_000)


@pytest.mark.parametrize("threads,chunk", [(1, 0), (4, 100_000)])
def test_damage_is_reported(tmp_path, threads, chunk):
    good = gzip.compress(FASTQ[:3_000_000], 6)
    src, dst = tmp_path / "bad.gz", tmp_path / "bad.out"
    for where in (len(good) // 3, len(good) // 2, len(good) - 6):  # data, data, stored crc
        bad = bytearray(good)
        bad[where] ^= 0x55
        src.write_bytes(bytes(bad))
        with pytest.raises(RuntimeError):
            gunzip(src, dst, threads, chunk)
    src.write_bytes(good[: len(good) // 2])  # truncated
    with pytest.raises(RuntimeError):
        gunzip(src, dst, threads, chunk)
    src.write_bytes(b"\x1f\x8b\x08")  # header cut short
    with pytest.raises(RuntimeError):
        gunzip(src, dst, threads, chunk)
    src.write_bytes(b"@r1\nACGT\n+\nIIII\n" * 10)  # not gzip at all
    with pytest.raises(RuntimeError):
        gunzip(src, dst, threads, chunk)


def test_reader_uses_the_parallel_decoder(tmp_path, monkeypatch):
    """nh_fastx_scan over a .gz goes through ByteSource: same digest with zlib and with 3 workers."""
    from tests.test_reader import scan
    data = open(os.path.join(os.path.dirname(__file__), "golden", "reads_se.fq"), "rb").read() * 40
    gz = tmp_path / "r.fq.gz"
    gz.write_bytes(gzip.compress(data, 6))
    monkeypatch.setenv("NOHUMAN_GZ_THREADS", "0")
    want = scan(gz)
    monkeypatch.setenv("NOHUMAN_GZ_THREADS", "3")
    monkeypatch.setenv("NOHUMAN_GZ_CHUNK", "50000")
    assert scan(gz) == want
    bad = bytearray(gz.read_bytes())
    bad[len(bad) // 2] ^= 0xFF
    gz.write_bytes(bytes(bad))
    with pytest.raises(RuntimeError):
        scan(gz)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crc_by_carryless_multiply_and_by_tables_agree(tmp_path):
    """The member CRCs are checked with a PCLMULQDQ CRC-32 where the CPU has it, slicing tables elsewhere
    (NOHUMAN_NO_CLMUL=1 forces the tables): both must accept what zlib wrote -- odd lengths, several members --
    and both must reject a flipped bit; the block-parallel gzip ENCODER uses the same routine, so what it writes
    must pass Python's gzip module."""
    import subprocess
    import sys
    rng = np.random.default_rng(3)
    data = b"".join(bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) + b"ACGT" * int(n) for n in (1, 63, 64, 65, 4097, 100001))
    gz = tmp_path / "m.gz"
    gz.write_bytes(gzip.compress(data[:70000], 6) + gzip.compress(data[70000:], 1))
    bad = bytearray(gz.read_bytes())
    bad[len(bad) // 3] ^= 0x10
    (tmp_path / "bad.gz").write_bytes(bytes(bad))
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from nohuman_amd import _lib\n"
            "L = _lib.lib()\n"
            "rc = L.nh_gunzip_file(sys.argv[1].encode(), sys.argv[2].encode(), 3, 65536, None)\n"
            "sys.exit(0 if rc == 0 else 3)\n") % ROOT
    for env_extra in ({}, {"NOHUMAN_NO_CLMUL": "1"}):
        env = dict(os.environ, **env_extra)
        out = tmp_path / "out.bin"
        assert subprocess.run([sys.executable, "-c", code, str(gz), str(out)], env=env).returncode == 0
        assert out.read_bytes() == data
        assert subprocess.run([sys.executable, "-c", code, str(tmp_path / "bad.gz"), str(out)], env=env).returncode == 3
        plain = tmp_path / "p.bin"
        plain.write_bytes(data)
        code2 = ("import sys; sys.path.insert(0, %r)\nfrom nohuman_amd import _lib\n"
                 "sys.exit(_lib.lib().nh_compress_file(sys.argv[1].encode(), sys.argv[2].encode(), 2, 3))\n") % ROOT
        assert subprocess.run([sys.executable, "-c", code2, str(plain), str(tmp_path / "enc.gz")], env=env).returncode == 0
        assert gzip.decompress((tmp_path / "enc.gz").read_bytes()) == data
