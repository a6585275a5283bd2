"""Sanitizers over the host-side code (SURVEY.md section 5 "Race detection / sanitizers").  GPU
AddressSanitizer is not available on the pool, and nh_run has some ten cooperating host threads: the product's host sources
nh_inflate.cpp (speculative multi-threaded gzip decoder), nh_fastx.cpp (block reader with its read-ahead thread) and
nh_codec.cpp (the host gzip encoder's worker pool) are built AS THEY ARE into tools/san_host.cpp with
-fsanitize=address,undefined and again with -fsanitize=thread, and run over gzip streams of every shape the decoder's own tests
use, at 1 / 4 / 8 threads; round 6: the same streams through RangeGunzip, the hybrid reader's host lane, cell by cell.  The CPU oracle (test infrastructure) runs its own test file once under ASan + UBSan.
CPU only."""
import gzip
import os
import shutil
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nohuman_amd", "csrc")
BAD = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer", "SUMMARY: UndefinedBehaviorSanitizer")


def _build(out, flags):
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + SRC, os.path.join(ROOT, "tools", "san_host.cpp"),
           os.path.join(ROOT, "tools", "san_stubs.cpp"), os.path.join(SRC, "nh_inflate.cpp"), os.path.join(SRC, "nh_fastx.cpp"),
           os.path.join(SRC, "nh_codec.cpp"), "-o", out, "-lz", "-lpthread", "-ldl", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("no g++ / ROCm runtime")
    d = tmp_path_factory.mktemp("san")
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:  # the two builds side by side (~20 s)
        a = ex.submit(_build, str(d / "san_asan"), ["-fsanitize=address,undefined"])
        t = ex.submit(_build, str(d / "san_tsan"), ["-fsanitize=thread"])
        return {"asan": a.result(), "tsan": t.result()}


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    from tests.test_codec import fastq_like
    from tests.test_gunzip import deflate_raw
    d = tmp_path_factory.mktemp("corpus")
    rng = np.random.default_rng(9)
    fq = fastq_like(2_500_000, seed=33)
    texts = {"fastq": fq, "zeros": bytes(1_500_000), "random": rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes(),
             "tiny": b"hello, world\n", "empty": b""}
    streams = {
        "fastq_l1": gzip.compress(fq, 1), "fastq_l6": gzip.compress(fq, 6), "fastq_l9": gzip.compress(fq, 9),
        "members": b"".join(gzip.compress(fq[i:i + 400_000], 6) for i in range(0, len(fq), 400_000)),
        "sync_flush": deflate_raw(fq, flush_every=70_000), "full_flush": deflate_raw(fq, flush_every=200_000, flush=zlib.Z_FULL_FLUSH),
        "huffman_only": deflate_raw(fq[:800_000], strategy=zlib.Z_HUFFMAN_ONLY), "fixed": deflate_raw(fq[:300_000], strategy=zlib.Z_FIXED),
        "zeros": gzip.compress(texts["zeros"], 6), "random": gzip.compress(texts["random"], 6),
        "tiny": gzip.compress(texts["tiny"]), "empty": gzip.compress(b""),
        "trailing_zeros": gzip.compress(fq[:500_000], 6) + bytes(3000),
    }
    good = gzip.compress(fq[:900_000], 6)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A
    streams["damaged"] = bytes(bad)        # (errors are fine: crashes, overruns and races are not)
    streams["truncated"] = good[: len(good) * 2 // 3]
    paths = {}
    for k, v in streams.items():
        p = d / (k + ".gz")
        p.write_bytes(v)
        paths[k] = str(p)
    plain = {}
    for k, v in texts.items():
        p = d / (k + ".txt")
        p.write_bytes(v)
        plain[k] = str(p)
    return paths, plain


def _run(exe, args, kind):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([exe] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert not any(b in out for b in BAD), "%s %s:\n%s" % (kind, args, out[-4000:])
    assert r.returncode == 0, "%s %s: rc %d\n%s" % (kind, args, r.returncode, out[-2000:])
    assert "MISMATCH" not in out
    return out


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_gzip_decoder_and_block_reader_under_sanitizers(binaries, corpus, kind):
    paths, _ = corpus
    for name, p in paths.items():
        for threads, chunk in ((1, 0), (4, 70_000), (8, 200_000)):
            if kind == "tsan" and name in ("fastq_l1", "fastq_l9", "huffman_only", "fixed") and threads == 1:
                continue  # (one thread: nothing to race; keeps the suite within its minute)
            _run(binaries[kind], ["gunzip", p, threads, chunk], kind)


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_range_decoder_of_the_hybrid_reader_under_sanitizers(binaries, corpus, kind):
    """RangeGunzip (round 6): a cell's chunks decoded ahead by a worker pool, stitched later from a position and window handed in --
    a second way through the decoder's shared state (tails, jobs, pieces), with its own thread hand-overs"""
    paths, _ = corpus
    for name, p in paths.items():
        if kind == "tsan" and name in ("fastq_l1", "fastq_l9", "huffman_only", "fixed", "zeros"):
            continue
        for threads, cell, chunk, every in ((4, 200_000, 40_000, 1), (8, 120_000, 9_000, 2)):
            out = _run(binaries[kind], ["ranges", p, threads, cell, chunk, every], kind)
            if name == "fastq_l6":
                assert "ranges ok" in out and "zlib ok" in out


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_gzip_encoder_under_sanitizers(binaries, corpus, kind):
    _, plain = corpus
    for name, p in plain.items():
        for threads in (1, 4, 8):
            out = _run(binaries[kind], ["gzip", p, threads], kind)
            assert "inflated back" in out


def test_oracle_under_asan_and_ubsan():
    """oracle/k2_oracle.c (the checker every parity test trusts) through its own test file, built with -fsanitize=address,undefined."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libk2oracle_asan.so"])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, K2ORACLE_LIB=os.path.join(ROOT, "oracle", "libk2oracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")  # (leaks: the interpreter's own)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert not any(b in out for b in BAD), out[-4000:]
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out
