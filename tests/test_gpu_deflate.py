"""gzip output encoded on the GPU (nohuman_amd/csrc/nh_deflate.hip; SURVEY.md 8f-4, the reference's stage is
compression.rs:214-233).  Parity target is the decompressed content (compression.rs:282-288): every stream is
inflated by zlib -- which also checks the member's CRC-32 and length -- and by this repo's own parallel reader."""
import ctypes as C
import gzip
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nohuman_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu


def gpu_gzip(data, path):
    L = _lib.lib()
    stats = (C.c_uint64 * 2)()
    buf = (C.c_char * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    rc = L.nh_gzip_gpu_file(0, buf, len(data), os.fsencode(path), stats)
    assert rc == 0, _lib.lib().nh_last_error().decode()
    assert os.path.getsize(path) == stats[0]
    return stats[0], stats[1]


def fastq_text(n_reads, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_reads):
        L = 150
        seq = "".join("ACGT"[c] for c in rng.integers(0, 4, L))
        q = np.where(rng.random(L) < 0.06, ord(":"), ord("F")).astype(np.uint8)
        cut = int(L * (0.3 + 0.7 * rng.random() ** 0.4))
        q[cut:] = rng.choice(np.frombuffer(b"F:,#", dtype=np.uint8), L - cut)
        out.append("@NH1:7:HGF2YDSXX:1:%d:%d:%d 1:N:0:GATTACAG\n%s\n+\n%s\n"
                   % (1101 + i // 5000, 10000 + int(rng.integers(0, 25000)), 10000 + (i * 17) // 10, seq,
                      q.tobytes().decode()))
    return "".join(out).encode()


def check(data, path):
    size, _ = gpu_gzip(data, path)
    raw = open(path, "rb").read()
    assert raw[:3] == b"\x1f\x8b\x08"
    assert gzip.decompress(raw) == data          # zlib: content, CRC-32, ISIZE
    d = zlib.decompressobj(31)                    # one member, nothing behind it
    assert d.decompress(raw) == data and d.eof and d.unused_data == b""
    return size


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 4095, 65535, 65536, 65537, 131072, 200001])
def test_sizes_around_step_block_and_region_edges(tmp_path, n):
    rng = np.random.default_rng(n)
    text = fastq_text(max(1, n // 300 + 1), n)
    data = (text * (n // len(text) + 1))[:n]
    check(data, str(tmp_path / "a.gz"))
    # and bytes with no structure at all: stored blocks
    check(rng.integers(0, 256, n, dtype=np.uint8).tobytes(), str(tmp_path / "b.gz"))


def test_fastq_text_ratio_and_own_reader(tmp_path):
    data = fastq_text(40000, 7)
    size = check(data, str(tmp_path / "a.gz"))
    z6 = len(zlib.compress(data, 6))
    print("GPU gzip %.3f : 1, zlib -6 %.3f : 1" % (len(data) / size, len(data) / z6))
    assert size < 1.12 * z6, "ratio fell behind zlib -6 by more than 12 %"
    # the repo's own reader (speculative parallel inflate) takes the stream as well
    out = str(tmp_path / "a.txt")
    st = (C.c_uint64 * 3)()
    rc = _lib.lib().nh_gunzip_file(os.fsencode(str(tmp_path / "a.gz")), os.fsencode(out), 4, 1 << 20, st)
    assert rc == 0, _lib.lib().nh_last_error().decode()
    assert open(out, "rb").read() == data


def test_runs_long_matches_and_every_byte_value(tmp_path):
    rng = np.random.default_rng(3)
    parts = [b"\0" * 100000, bytes(range(256)) * 300, b"A" * 70000, b"ACGT" * 20000,
             rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(), b"F" * 258, b"F" * 259, b"xyz" * 5,
             rng.integers(0, 4, 90000, dtype=np.uint8).tobytes()]
    check(b"".join(parts), str(tmp_path / "a.gz"))
    # a block whose symbols force long codes: geometric counts
    geo = b"".join(bytes([i]) * (1 << min(i, 14)) for i in range(24))
    check(geo + bytes(rng.permutation(np.frombuffer(geo, dtype=np.uint8))), str(tmp_path / "b.gz"))


def test_several_chunks(tmp_path, monkeypatch):
    """More than one chunk (64 MiB here, 128 MiB by default: NOHUMAN_GZIP_CHUNK_MB is read when an encoder is set up): two
    buffers in flight, the prices handed from chunk to chunk."""
    monkeypatch.setenv("NOHUMAN_GZIP_CHUNK_MB", "64")
    unit = fastq_text(20000, 11)
    data = unit * (150_000_000 // len(unit))
    size = check(data, str(tmp_path / "a.gz"))
    assert size < len(data) / 3


@pytest.mark.parametrize("batch_frags", [700, 4000, 50000])
def test_run_takes_long_spans_from_the_device_copy_of_the_batch(tmp_path, monkeypatch, batch_frags):
    """nh_run with gzip outputs: kept records in spans of 32 KiB and more are compressed from the copy of the batch's
    text that the classifier worked on (no second trip over PCIe), shorter spans and reformatted records are
    staged -- the two mixed in one stream, several batches, both mate files.  Decompressed bytes == plain outputs,
    for the GPU encoder and for the host encoder (NOHUMAN_GZIP=host)."""
    from nohuman_amd import Engine
    gold = os.path.join(os.path.dirname(__file__), "golden")
    rng = np.random.default_rng(5)
    files = []
    for m in ("1", "2"):
        fixture = open(os.path.join(gold, "reads_pe_%s.fq" % m), "rb").read()   # human and other reads, mixed
        parts = []
        for block in range(6):
            parts.append(fixture * 3)                                            # short spans between removed reads
            for i in range(1500):                                                # a long run of kept reads
                seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150))
                parts.append(b"@r%d_%d/%s\n%s\n+\n%s\n" % (block, i, m.encode(), seq, b"F" * 150))
        p = tmp_path / ("in_%s.fq" % m)
        p.write_bytes(b"".join(parts))
        files.append(str(p))
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", str(batch_frags))
    with Engine.open(os.path.join(gold, "toy_db")) as eng:
        st = eng.run(files[0], str(tmp_path / "p_1.fq"), in2=files[1], out2=str(tmp_path / "p_2.fq"))
        eng.run(files[0], str(tmp_path / "g_1.gz"), in2=files[1], out2=str(tmp_path / "g_2.gz"), out_codec=2)
        monkeypatch.setenv("NOHUMAN_GZIP", "host")
        eng.run(files[0], str(tmp_path / "h_1.gz"), in2=files[1], out2=str(tmp_path / "h_2.gz"), out_codec=2, codec_threads=2)
    assert 0 < st.classified < st.total_sequences
    for m in ("1", "2"):
        want = (tmp_path / ("p_%s.fq" % m)).read_bytes()
        assert len(want) > 2_000_000
        assert gzip.decompress((tmp_path / ("g_%s.gz" % m)).read_bytes()) == want
        assert gzip.decompress((tmp_path / ("h_%s.gz" % m)).read_bytes()) == want


def test_encoder_with_pageable_staging_buffers(tmp_path):
    """A host that will not page-lock the encoder's staging buffers gets pageable ones (NOHUMAN_NO_PINNED forces it;
    read once per process, hence the child): same stream contents."""
    import subprocess
    data = fastq_text(3000, 21)
    src = tmp_path / "in.fq"
    src.write_bytes(data)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from nohuman_amd import _lib\n"
            "d = open(sys.argv[1], 'rb').read()\n"
            "buf = (C.c_char * len(d)).from_buffer_copy(d)\n"
            "rc = _lib.lib().nh_gzip_gpu_file(0, buf, len(d), sys.argv[2].encode(), None)\n"
            "sys.exit(rc)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code, str(src), str(tmp_path / "o.gz")],
                       env=dict(os.environ, NOHUMAN_NO_PINNED="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert gzip.decompress((tmp_path / "o.gz").read_bytes()) == data


def test_system_gzip_accepts_the_stream(tmp_path):
    """Interoperability: the system's gzip tool tests and expands the member."""
    import shutil
    import subprocess
    if not shutil.which("gzip"):
        pytest.skip("no gzip tool on this box")
    data = fastq_text(5000, 22)
    gpu_gzip(data, str(tmp_path / "a.gz"))
    assert subprocess.run(["gzip", "-t", str(tmp_path / "a.gz")]).returncode == 0
    out = subprocess.run(["gzip", "-dc", str(tmp_path / "a.gz")], capture_output=True)
    assert out.returncode == 0 and out.stdout == data


def test_compress_file_on_the_device(tmp_path):
    """nh_compress_file_device: the reference's compress stage (file -> file) with the gzip case on the GPU."""
    data = fastq_text(8000, 23)
    src = tmp_path / "kraken_out.fq"
    src.write_bytes(data)
    L = _lib.lib()
    assert L.nh_compress_file_device(os.fsencode(str(src)), os.fsencode(str(tmp_path / "o.gz")), 2, 4, 0) == 0
    assert gzip.decompress((tmp_path / "o.gz").read_bytes()) == data
    assert L.nh_compress_file_device(os.fsencode(str(src)), os.fsencode(str(tmp_path / "o.zst")), 4, 2, 0) == 0   # others: as on the host
    assert (tmp_path / "o.zst").read_bytes()[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD])


@pytest.mark.parametrize("period", [32767, 32768, 32769, 40000, 65535])
def test_repeats_at_and_beyond_the_window(tmp_path, period):
    """A match may reach back 32768 bytes and not one more: blocks of random bytes repeated at distances around
    the window (inside one region and across regions)."""
    rng = np.random.default_rng(period)
    unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
    check(unit * 5 + unit[:1000], str(tmp_path / "a.gz"))


def _mixture(seed):
    """Segments of many kinds glued together: what a generic encoder must survive."""
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(int(rng.integers(3, 12))):
        kind = int(rng.integers(0, 8))
        n = int(rng.integers(1, 60000))
        if kind == 0:
            parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif kind == 1:
            parts.append(bytes([int(rng.integers(0, 256))]) * n)
        elif kind == 2:
            k = int(rng.integers(2, 17))
            parts.append(rng.integers(0, k, n, dtype=np.uint8).tobytes())
        elif kind == 3:
            unit = rng.integers(0, 256, int(rng.integers(1, 700)), dtype=np.uint8).tobytes()
            parts.append((unit * (n // len(unit) + 1))[:n])
        elif kind == 4:
            parts.append(fastq_text(n // 350 + 1, seed + n)[:n])
        elif kind == 5:
            p = np.cumsum(rng.random(256) ** 8)  # a very skewed alphabet
            parts.append(np.searchsorted(p / p[-1], rng.random(n)).astype(np.uint8).tobytes())
        elif kind == 6 and parts:
            prev = b"".join(parts)
            a = int(rng.integers(0, len(prev)))
            parts.append(prev[a:a + n])                     # an old stretch again: long matches at any distance
        else:
            parts.append(b"".join(b"%d\t%d\n" % (i, i * i) for i in range(n // 12 + 1)))
    return b"".join(parts)


@pytest.mark.parametrize("seed", range(int(os.environ.get("NH_DEFLATE_SEEDS", "24"))))  # (a soak run sets more)
def test_random_mixtures(tmp_path, seed):
    check(_mixture(seed), str(tmp_path / "a.gz"))


def test_verify_mode_inflates_every_chunk_on_the_host(tmp_path):
    """NOHUMAN_GZIP_VERIFY=1: every chunk's stream is inflated again by zlib and its length and CRC-32 compared with
    the text's before it is written (read once per process, hence the child)."""
    import subprocess
    data = _mixture(101) + fastq_text(4000, 9) + _mixture(102)
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from nohuman_amd import _lib\n"
            "d = open(sys.argv[1], 'rb').read()\n"
            "buf = (C.c_char * len(d)).from_buffer_copy(d)\n"
            "rc = _lib.lib().nh_gzip_gpu_file(0, buf, len(d), sys.argv[2].encode(), None)\n"
            "print(_lib.lib().nh_last_error().decode() if rc else 'ok')\n"
            "sys.exit(rc != 0)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code, str(src), str(tmp_path / "o.gz")],
                       env=dict(os.environ, NOHUMAN_GZIP_VERIFY="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert gzip.decompress((tmp_path / "o.gz").read_bytes()) == data


def _reads_with_qualities(kind, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = 150
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L))
        if kind == "forty":      # forty quality values: a random walk around a falling mean
            base = 38 - np.linspace(0, 12, L) * rng.random()
            q = np.clip(base + rng.normal(0, 3, L) + np.cumsum(rng.normal(0, 0.6, L)), 2, 41).astype(np.uint8) + 33
        elif kind == "spread":   # long-read-like: a wide spread, no structure
            q = np.clip(rng.normal(20, 7, L), 1, 50).astype(np.uint8) + 33
        else:                    # eight bins
            bins = np.array([2, 6, 15, 22, 27, 33, 37, 40], dtype=np.uint8)
            idx = np.clip((7 - np.linspace(0, 3, L) * rng.random() + rng.normal(0, 0.8, L)).round(), 0, 7).astype(int)
            q = bins[idx] + 33
        out.append(b"@SRR1234567.%d %d/1\n%s\n+\n%s\n" % (i + 1, i + 1, seq, q.tobytes()))
    return b"".join(out)


@pytest.mark.parametrize("kind,slack", [("forty", 1.05), ("spread", 1.03), ("bins", 1.12)])
def test_starting_prices_are_chosen_per_stream(tmp_path, kind, slack):
    """The encoder tries both starting price sets on a stream's first regions (bases literal and cheap / zlib-like)
    and goes on with the smaller: texts whose qualities have many values want the second.  Sizes against zlib -6."""
    data = _reads_with_qualities(kind, 12000, 31)
    size = check(data, str(tmp_path / "a.gz"))
    z6 = len(zlib.compress(data, 6))
    print("%s: GPU %.3f : 1, zlib -6 %.3f : 1" % (kind, len(data) / size, len(data) / z6))
    assert size < slack * z6
