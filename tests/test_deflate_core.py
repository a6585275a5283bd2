"""The format logic of the GPU gzip encoder on a CPU: tools/deflate_model.cpp drives the SAME lane-level and
sequential functions as the kernel (nohuman_amd/csrc/nh_deflate_core.h: symbol mapping, match finder, code
construction with the length limit, run-length coded block headers, stored blocks) with a 64-lane loop.  Its
streams must inflate, under zlib, to the input -- member CRC and length included."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dfl") / "deflate_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "nohuman_amd", "csrc"),
                           os.path.join(ROOT, "tools", "deflate_model.cpp"), "-o", exe, "-lz"])
    return exe


def run_model(model, data, tmp_path, *args):
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.gz")
    open(src, "wb").write(data)
    subprocess.check_call([model, src, dst] + [str(a) for a in args], stderr=subprocess.DEVNULL)
    raw = open(dst, "rb").read()
    d = zlib.decompressobj(31)
    assert d.decompress(raw) == data and d.eof and d.unused_data == b""
    return len(raw)


def fastq(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150))
        q = np.where(rng.random(150) < 0.06, ord(":"), ord("F")).astype(np.uint8)
        cut = int(150 * (0.3 + 0.7 * rng.random() ** 0.4))
        q[cut:] = rng.choice(np.frombuffer(b"F:,#", dtype=np.uint8), 150 - cut)
        out.append(b"@NH1:7:HGF2YDSXX:1:%d:%d:%d 1:N:0:GATTACAG\n%s\n+\n%s\n"
                   % (1101 + i // 5000, 10000 + int(rng.integers(0, 25000)), 10000 + (i * 17) // 10, seq, q.tobytes()))
    return b"".join(out)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 65, 258, 259, 300, 4096, 65535, 65536, 65537, 150000])
def test_model_streams_inflate_to_the_input(model, tmp_path, n):
    text = fastq(n // 300 + 2, n)
    run_model(model, (text * (n // len(text) + 1))[:n], tmp_path)
    run_model(model, np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8).tobytes(), tmp_path)   # stored blocks


def test_model_on_runs_every_byte_value_and_skewed_counts(model, tmp_path):
    rng = np.random.default_rng(1)
    parts = [b"\0" * 70000, bytes(range(256)) * 100, b"ACGT" * 9000, b"F" * 258, b"F" * 259,
             rng.integers(0, 4, 40000, dtype=np.uint8).tobytes()]
    run_model(model, b"".join(parts), tmp_path)
    # counts 1, 2, 4, ... force codes longer than 15 bits before the limit is applied
    geo = b"".join(bytes([i]) * (1 << min(i, 14)) for i in range(24))
    run_model(model, geo + bytes(rng.permutation(np.frombuffer(geo, dtype=np.uint8))), tmp_path)
    # small regions and blocks: many headers, every region boundary an empty stored block
    run_model(model, fastq(600, 2), tmp_path, 4096, 1024)


def test_model_ratio_on_fastq_text_is_near_zlib_6(model, tmp_path):
    data = fastq(12000, 3)
    size = run_model(model, data, tmp_path)
    z6 = len(zlib.compress(data, 6))
    assert size < 1.10 * z6, (size, z6)
    assert gzip.decompress(open(str(tmp_path / "out.gz"), "rb").read()) == data


@pytest.mark.parametrize("period", [32767, 32768, 32769, 40000, 65535])
def test_model_repeats_at_and_beyond_the_window(model, tmp_path, period):
    unit = np.random.default_rng(period).integers(0, 256, period, dtype=np.uint8).tobytes()
    run_model(model, unit * 5 + unit[:1000], tmp_path)
