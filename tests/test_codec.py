"""Output compression stage (nh_compress_file) without a GPU -- content parity with
CompressionFormat::compress (/root/reference/src/compression.rs:182-268; the reference's own tests
at compression.rs:504-586 pin library byte streams, ours pin the decompressed content and the
container magic of compression.rs:282-288)."""
import bz2
import gzip
import lzma
import os
import zlib

import numpy as np
import pytest

from nohuman_amd import _lib

NONE, BZIP2, GZIP, XZ, ZSTD = 0, 1, 2, 3, 4


def compress(src, dst, codec, threads):
    L = _lib.lib()
    rc = L.nh_compress_file(os.fsencode(src), os.fsencode(dst), codec, threads)
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())


def fastq_like(n_bytes, seed=1):
    rng = np.random.default_rng(seed)
    out = []
    size = 0
    i = 0
    while size < n_bytes:
        seq = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 150)])
        qual = bytes((rng.integers(0, 12, 150) + 58).astype(np.uint8))
        rec = b"@read.%d some description\n%s\n+\n%s\n" % (i, seq, qual)
        out.append(rec)
        size += len(rec)
        i += 1
    return b"".join(out)[:n_bytes]


BLOCK = 512 * 1024


@pytest.mark.parametrize("size", [0, 1, 1000, BLOCK - 1, BLOCK, BLOCK + 1, 3 * BLOCK + 12345])
@pytest.mark.parametrize("threads", [1, 3])
def test_gzip_blocks_round_trip_as_one_member(tmp_path, size, threads):
    data = fastq_like(size, seed=size % 97)
    src, dst = tmp_path / "in.fq", tmp_path / "out.fq.gz"
    src.write_bytes(data)
    compress(src, dst, GZIP, threads)
    raw = dst.read_bytes()
    assert raw[:2] == b"\x1f\x8b"  # compression.rs:283
    d = zlib.decompressobj(31)  # exactly one gzip member: nothing may follow its trailer
    assert d.decompress(raw) == data
    assert d.eof and d.unused_data == b""
    assert gzip.decompress(raw) == data


def test_gzip_many_threads_and_ratio(tmp_path):
    data = fastq_like(6 * BLOCK + 777, seed=5)
    src = tmp_path / "in.fq"
    src.write_bytes(data)
    sizes = {}
    for threads in (1, 8):
        dst = tmp_path / ("out%d.gz" % threads)
        compress(src, dst, GZIP, threads)
        assert gzip.decompress(dst.read_bytes()) == data
        sizes[threads] = dst.stat().st_size
    assert sizes[1] == sizes[8]  # block cutting does not depend on the worker count
    ref = len(zlib.compress(data, 6))
    assert sizes[1] < 1.03 * ref  # the 32 KiB dictionary carry keeps the block-parallel ratio close


def test_gzip_binary_and_incompressible(tmp_path):
    data = np.random.default_rng(3).integers(0, 256, 2 * BLOCK + 99, dtype=np.uint8).tobytes()
    src, dst = tmp_path / "in.bin", tmp_path / "out.gz"
    src.write_bytes(data)
    compress(src, dst, GZIP, 4)
    assert gzip.decompress(dst.read_bytes()) == data


def _zstd_decompress(raw, size):
    """The image has libzstd.so.1 but no Python binding: decompress through ctypes."""
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    dst = ctypes.create_string_buffer(size + 16)
    n = z.ZSTD_decompress(dst, size + 16, raw, len(raw))
    assert not z.ZSTD_isError(n)
    return dst.raw[:n]


def test_other_codecs(tmp_path):
    data = fastq_like(200_000, seed=9)
    src = tmp_path / "in.fq"
    src.write_bytes(data)
    compress(src, tmp_path / "o.fq", NONE, 1)
    assert (tmp_path / "o.fq").read_bytes() == data
    compress(src, tmp_path / "o.bz2", BZIP2, 1)
    raw = (tmp_path / "o.bz2").read_bytes()
    assert raw[:2] == b"\x42\x5a" and bz2.decompress(raw) == data
    compress(src, tmp_path / "o.xz", XZ, 2)
    raw = (tmp_path / "o.xz").read_bytes()
    assert raw[:5] == b"\xfd\x37\x7a\x58\x5a" and lzma.decompress(raw) == data
    for threads in (1, 4):
        compress(src, tmp_path / "o.zst", ZSTD, threads)
        raw = (tmp_path / "o.zst").read_bytes()
        assert raw[:4] == b"\x28\xb5\x2f\xfd"  # compression.rs:285
        assert _zstd_decompress(raw, len(data)) == data
        assert raw[4] & 0x04  # frame header descriptor: content checksum present (include_checksum(true))


def test_errors(tmp_path):
    with pytest.raises(RuntimeError):
        compress(tmp_path / "missing", tmp_path / "o.gz", GZIP, 1)
    src = tmp_path / "in"
    src.write_bytes(b"x")
    with pytest.raises(RuntimeError):
        compress(src, tmp_path / "no_such_dir" / "o.gz", GZIP, 1)
    with pytest.raises(RuntimeError):
        compress(src, tmp_path / "o", 17, 1)
