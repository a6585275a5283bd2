"""The C++ `nohuman` host (nohuman_amd/bin/nohuman) against the reference CLI contract
(/root/reference/src/main.rs:21-386): flags, error texts / exit codes, default output names, codec
decision, temp-dir clean-up, log lines.  GPU-free checks first, full runs marked gpu."""
import gzip
import json
import os
import shutil
import subprocess

import pytest

from tests.fastq_util import read_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "nohuman_amd", "bin", "nohuman")
GOLD = os.path.join(ROOT, "tests", "golden")
DB = os.path.join(GOLD, "toy_db")

pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="CLI host not built")


def run(args, cwd=None, env=None):
    e = dict(os.environ)
    e.pop("NOHUMAN_DB", None)
    if env:
        e.update(env)
    return subprocess.run([BIN] + args, cwd=cwd, env=e, capture_output=True, text=True)


def test_help_lists_the_reference_flags():
    r = run(["--help"])
    assert r.returncode == 0
    for flag in ("--out1", "--out2", "--check", "--download", "--db", "--db-version", "--list-db-versions",
                 "--output-type", "--threads", "--human", "--conf", "--kraken-output", "--kraken-report",
                 "--verbose"):
        assert flag in r.stdout


def test_argument_errors_exit_2(tmp_path):
    f = tmp_path / "in.fq"
    f.write_text("@r\nACGT\n+\nIIII\n")
    assert run([]).returncode == 2  # INPUT required unless --check/--download/--list-db-versions
    r = run(["-t", "0", str(f)])
    assert r.returncode == 2 and "--threads" in r.stderr
    r = run(["-C", "1.1", str(f)])
    assert r.returncode == 2 and "Confidence score must be in the closed interval [0, 1]" in r.stderr
    r = run(["-C", "abc", str(f)])
    assert r.returncode == 2 and "Confidence score must be a number" in r.stderr
    r = run(["-F", "q", str(f)])
    assert r.returncode == 2 and "is not a valid output format" in r.stderr
    r = run([str(tmp_path / "missing.fq")])
    assert r.returncode == 2 and "does not exist" in r.stderr


def _has_gpu():
    return run(["-c"]).returncode == 0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_check_fails_loudly_without_a_device():
    r = run(["--check"])
    assert r.returncode == 1
    assert "The following dependencies are missing:" in r.stderr
    assert "Error: Missing dependencies" in r.stderr


@pytest.mark.gpu
def test_check_ok():
    r = run(["-c"])
    assert r.returncode == 0
    assert r.stderr.rstrip().endswith("INFO ] All dependencies are available")


@pytest.mark.gpu
def test_single_end_default_name_and_codec_from_input(tmp_path):
    """no -o: `<stem>.nohuman.fq` beside the input, compressed like the input (main.rs:238-245,274-290)."""
    exp = json.load(open(os.path.join(GOLD, "expected_se.json")))
    calls = [r["by_conf"]["0.0"][0] for r in exp["records"]]
    reads = read_fastq(os.path.join(GOLD, "reads_se.fq"))
    inp = tmp_path / "sample.fq.gz"
    with open(os.path.join(GOLD, "reads_se.fq"), "rb") as f, gzip.open(inp, "wb") as g:
        g.write(f.read())
    r = run(["-D", DB, "-t", "2", str(inp)], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "sample.nohuman.fq.gz"
    assert out.exists()
    want = b"".join(h + b"\n" + s + b"\n+\n" + q + b"\n" for (h, _i, s, q), c in zip(reads, calls) if not c)
    assert gzip.open(out, "rb").read() == want
    n_class = sum(1 for c in calls if c)
    assert "%d / %d (%.2f%%) sequences classified as human; %d (%.2f%%) as non-human" % (
        n_class, len(calls), 100.0 * n_class / len(calls), len(calls) - n_class,
        100.0 * (len(calls) - n_class) / len(calls)) in r.stderr
    assert "Removing human reads..." in r.stderr and "Done." in r.stderr
    assert not [d for d in os.listdir(tmp_path) if d.startswith("nohuman")]  # temp dir removed


@pytest.mark.gpu
def test_paired_keep_human_with_outputs_report_and_kraken_file(tmp_path):
    exp = json.load(open(os.path.join(GOLD, "expected_pe.json")))
    ext = exp["meta"]["external_ids"]
    calls = [r["by_conf"]["0.1"][0] for r in exp["records"]]
    r1 = read_fastq(os.path.join(GOLD, "reads_pe_1.fq"))
    r2 = read_fastq(os.path.join(GOLD, "reads_pe_2.fq"))
    for n in ("reads_pe_1.fq", "reads_pe_2.fq"):
        shutil.copy(os.path.join(GOLD, n), tmp_path / n)
    o1, o2 = tmp_path / "h_1.fq", tmp_path / "h_2.fq.gz"
    r = run(["--db", DB, "-H", "-C", "0.1", "-t", "4", "-o", str(o1), "-O", str(o2), "-k", str(tmp_path / "k.txt"),
             "-r", str(tmp_path / "rep.txt"), "reads_pe_1.fq", "reads_pe_2.fq"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Keeping human reads..." in r.stderr

    def want(reads):
        return b"".join(h + b" kraken:taxid|%d" % ext[c] + b"\n" + s + b"\n+\n" + q + b"\n"
                        for (h, _i, s, q), c in zip(reads, calls) if c)
    # codec follows --out1's extension (none) for BOTH outputs, as in the reference (main.rs:240-241)
    assert o1.read_bytes() == want(r1)
    assert o2.read_bytes() == want(r2)
    assert len((tmp_path / "k.txt").read_text().splitlines()) == len(calls)
    assert (tmp_path / "rep.txt").read_text().count("\n") >= 5
    assert "Kraken output file written to:" in r.stderr and "Kraken report file written to:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flag,ext,magic", [("z", "zst", b"\x28\xb5\x2f\xfd"), ("b", "bz2", b"BZh"),
                                            ("x", "xz", b"\xfd7zXZ"), ("g", "gz", b"\x1f\x8b"), ("u", "", b"@")])
def test_output_type_flag_selects_the_codec(tmp_path, flag, ext, magic):
    """-F overrides the codec guessed from the input / output name (main.rs:238-245); the file gets
    the codec's extension appended (compression.rs:107-118) and the container magic of
    compression.rs:282-288; the content is the kept reads."""
    import bz2
    import lzma
    from tests.test_codec import _zstd_decompress
    exp = json.load(open(os.path.join(GOLD, "expected_se.json")))
    calls = [r["by_conf"]["0.0"][0] for r in exp["records"]]
    reads = read_fastq(os.path.join(GOLD, "reads_se.fq"))
    shutil.copy(os.path.join(GOLD, "reads_se.fq"), tmp_path / "s.fq")
    r = run(["-D", DB, "-t", "3", "-F", flag, "s.fq"], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    out = tmp_path / ("s.nohuman.fq" + ("." + ext if ext else ""))
    assert out.exists(), os.listdir(tmp_path)
    raw = out.read_bytes()
    assert raw.startswith(magic)
    want = b"".join(h + b"\n" + s + b"\n+\n" + q + b"\n" for (h, _i, s, q), c in zip(reads, calls) if not c)
    got = {"z": lambda: _zstd_decompress(raw, len(want)), "b": lambda: bz2.decompress(raw),
           "x": lambda: lzma.decompress(raw), "g": lambda: gzip.decompress(raw), "u": lambda: raw}[flag]()
    assert got == want


@pytest.mark.gpu
def test_database_resolution_errors_and_env(tmp_path):
    f = tmp_path / "in.fq"
    f.write_text("@r\nACGT\n+\nIIII\n")
    r = run(["-D", str(tmp_path / "nodb"), str(f)], cwd=tmp_path)
    assert r.returncode == 1
    assert "Database does not exist at" in r.stderr and "Run `nohuman --download` to fetch one." in r.stderr
    r = run(["--db-version", "HPRC.r9", "-D", str(tmp_path), str(f)], cwd=tmp_path)
    assert r.returncode == 1 and "Database version 'HPRC.r9' is not installed under" in r.stderr
    # versioned install discovered through nohuman-db.toml; NOHUMAN_DB supplies the root
    root = tmp_path / "dbroot"
    shutil.copytree(DB, root / "HPRC.rX" / "db")
    (root / "HPRC.rX" / "nohuman-db.toml").write_text('version = "HPRC.rX"\nadded = "2025-11-19"\n')
    r = run([str(f)], cwd=tmp_path, env={"NOHUMAN_DB": str(root)})
    assert r.returncode == 0, r.stderr
    assert "Using database version HPRC.rX at" in r.stderr
    assert (tmp_path / "in.nohuman.fq").exists()


@pytest.mark.gpu
def test_no_temporary_files_kept_reads_stream_into_the_encoder(tmp_path):
    """the reference has kraken2 write kraken_out*.fq into a temp dir "nohuman*" in the
    current directory and compresses them afterwards (main.rs:248-257,342-368).  Here the writer feeds the
    encoder directly: the run works from a current directory nothing can be created in, and leaves
    nothing but the outputs."""
    exp = json.load(open(os.path.join(GOLD, "expected_pe.json")))
    calls = [r["by_conf"]["0.0"][0] for r in exp["records"]]
    r1 = read_fastq(os.path.join(GOLD, "reads_pe_1.fq"))
    r2 = read_fastq(os.path.join(GOLD, "reads_pe_2.fq"))
    work = tmp_path / "readonly_cwd"
    work.mkdir()
    outd = tmp_path / "out"
    outd.mkdir()
    os.chmod(work, 0o555)
    try:
        r = run(["-D", DB, "-t", "4", "-o", str(outd / "a_1.fq.gz"), "-O", str(outd / "a_2.fq.gz"),
                 os.path.join(GOLD, "reads_pe_1.fq"), os.path.join(GOLD, "reads_pe_2.fq")], cwd=work)
    finally:
        os.chmod(work, 0o755)
    assert r.returncode == 0, r.stderr
    assert os.listdir(work) == []
    assert sorted(os.listdir(outd)) == ["a_1.fq.gz", "a_2.fq.gz"]
    for path, reads in ((outd / "a_1.fq.gz", r1), (outd / "a_2.fq.gz", r2)):
        want = b"".join(h + b"\n" + s + b"\n+\n" + q + b"\n" for (h, _i, s, q), c in zip(reads, calls) if not c)
        assert gzip.open(path, "rb").read() == want


@pytest.mark.gpu
def test_failed_run_leaves_nothing_behind(tmp_path):
    """a failing run must not leave a temp dir or half-written outputs in the cwd."""
    bad = tmp_path / "bad.fq.gz"
    with open(os.path.join(GOLD, "reads_se.fq"), "rb") as f:
        data = gzip.compress(f.read())
    bad.write_bytes(data[:-40])  # truncated member: the reader reports it after most reads went through
    r = run(["-D", DB, str(bad)], cwd=tmp_path)
    assert r.returncode == 1 and "Failed to run kraken2" in r.stderr
    assert sorted(os.listdir(tmp_path)) == ["bad.fq.gz"]


@pytest.mark.gpu
def test_failed_run_keeps_an_earlier_output_and_output_may_be_the_input(tmp_path):
    """the host stages its outputs ("<out>.partial", renamed on success).  A run that fails --
    here on a damaged input, and before that on a database that does not load -- must leave a file that
    already sits at the output path untouched; `-o` naming the input itself must not empty the input
    before it is read (the reference compresses its temporary kraken_out.fq to the output path last)."""
    good = os.path.join(GOLD, "reads_se.fq")
    out = tmp_path / "kept.fq"
    out.write_text("result of an earlier run\n")
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(gzip.compress(open(good, "rb").read())[:-40])
    r = run(["-D", DB, "-o", str(out), str(bad)], cwd=tmp_path)
    assert r.returncode == 1
    assert out.read_text() == "result of an earlier run\n"
    nodb = tmp_path / "nodb"
    nodb.mkdir()
    for n in ("hash.k2d", "opts.k2d", "taxo.k2d"):
        (nodb / n).write_bytes(b"")
    r = run(["-D", str(nodb), "-o", str(out), good], cwd=tmp_path)
    assert r.returncode == 1
    assert out.read_text() == "result of an earlier run\n"
    assert sorted(os.listdir(tmp_path)) == ["bad.fq.gz", "kept.fq", "nodb"]
    # output == input
    exp = json.load(open(os.path.join(GOLD, "expected_se.json")))
    calls = [rec["by_conf"]["0.0"][0] for rec in exp["records"]]
    inout = tmp_path / "inout.fq"
    shutil.copy(good, inout)
    r = run(["-D", DB, "-o", str(inout), str(inout)], cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    want = b"".join(h + b"\n" + s + b"\n+\n" + q + b"\n" for (h, _i, s, q), c in zip(read_fastq(good), calls) if not c)
    assert inout.read_bytes() == want and len(want) > 1000
    assert not (tmp_path / "inout.fq.partial").exists()
