"""Host-side mirror of the reference boundary; these tests restate the reference's own unit tests
for that boundary (/root/reference/src/lib.rs:153-222) plus the untested parse_kraken_stderr
(src/lib.rs:61-97) against the stderr grammar of SURVEY.md A.8.  No GPU."""
import pytest

from nohuman_amd import runner
from nohuman_amd.runner import CommandRunner, parse_confidence_score, parse_kraken_stderr

STDERR = """Loading database information... done.
1000 sequences (0.15 Mbp) processed in 0.043s (1395.3 Kseq/m, 209.30 Mbp/m).
  250 sequences classified (25.00%)
  750 sequences unclassified (75.00%)
"""


def test_parse_kraken_stderr_grammar():
    assert parse_kraken_stderr(STDERR) == (1000, 250, 750)


def test_parse_kraken_stderr_commas_and_missing_lines():
    assert parse_kraken_stderr("1,234,567 sequences (1 Mbp) processed in 1s\n") == (1234567, 0, 0)
    assert parse_kraken_stderr("") == (0, 0, 0)


def test_parse_kraken_stderr_bad_integer_is_an_error():
    with pytest.raises(ValueError):
        parse_kraken_stderr("abc sequences classified (1%)\n")


def test_check_path_exists(tmp_path):
    # src/lib.rs:189-200
    assert runner.check_path_exists(tmp_path) == tmp_path
    with pytest.raises(ValueError) as ei:
        runner.check_path_exists("fake.path")
    assert "does not exist" in str(ei.value)


def test_parse_confidence_score():
    # src/lib.rs:202-221
    assert parse_confidence_score("0.5") == 0.5
    assert parse_confidence_score("0") == 0.0
    assert parse_confidence_score("1") == 1.0
    for bad in ("1.1", "-0.1"):
        with pytest.raises(ValueError) as ei:
            parse_confidence_score(bad)
        assert str(ei.value) == "Confidence score must be in the closed interval [0, 1]"
    with pytest.raises(ValueError) as ei:
        parse_confidence_score("abc")
    assert str(ei.value) == "Confidence score must be a number"


def test_validate_db_directory(tmp_path):
    # src/lib.rs:119-141 and the empty-file convention of src/download.rs:507-549
    d = tmp_path / "db1"
    d.mkdir()
    for f in runner.REQUIRED_DB_FILES:
        (d / f).write_bytes(b"")
    assert runner.validate_db_directory(d) == d
    outer = tmp_path / "outer"
    (outer / "db").mkdir(parents=True)
    for f in runner.REQUIRED_DB_FILES:
        (outer / "db" / f).write_bytes(b"")
    assert runner.validate_db_directory(outer) == outer / "db"
    with pytest.raises(ValueError) as ei:
        runner.validate_db_directory(tmp_path / "nothing")
    assert "Required files (hash.k2d, opts.k2d, taxo.k2d) not found" in str(ei.value)


def test_argv_contract_of_main_rs():
    # the argv nohuman builds at src/main.rs:215-267
    argv = ["--threads", "4", "--db", "/db", "--output", "/dev/null", "--confidence", "0.1",
            "--report", "r.txt", "--paired", "--unclassified-out", "tmp/kraken_out#.fq",
            "a_1.fq", "a_2.fq"]
    o = CommandRunner.parse_argv(argv)
    assert o["threads"] == 4 and o["db"] == "/db" and o["output"] == "/dev/null"
    assert o["confidence"] == 0.1 and o["report"] == "r.txt" and o["paired"] is True
    assert o["unclassified_out"] == "tmp/kraken_out#.fq" and o["classified_out"] is None
    assert o["inputs"] == ["a_1.fq", "a_2.fq"]


def test_run_failure_maps_to_reference_error_text(tmp_path):
    # src/lib.rs:26-31: non-zero exit -> io::Error "<command> failed with stderr <text>"
    r = CommandRunner("kraken2")
    with pytest.raises(OSError) as ei:
        r.run(["--db", str(tmp_path), "--unclassified-out", str(tmp_path / "o.fq"),
               str(tmp_path / "missing.fq")])
    assert str(ei.value).startswith("kraken2 failed with stderr ")


def _fake_kraken2(tmp_path, body):
    exe = tmp_path / "kraken2"
    exe.write_text("#!/bin/sh\n" + body)
    exe.chmod(0o755)
    return str(exe)


def test_stock_subprocess_path_config0(tmp_path, monkeypatch, caplog):
    """NOHUMAN_STOCK_KRAKEN2=1: CommandRunner.run is the reference's own spawn-and-scrape path
    (src/lib.rs:22-48; BASELINE.json configs[0]).  A stand-in script plays kraken2: it must receive the
    argv verbatim and its stderr summary must come back as the three integers."""
    import logging
    args_seen = tmp_path / "argv.txt"
    exe = _fake_kraken2(tmp_path, 'printf "%%s\\n" "$@" > %s\ncat >&2 <<X\n%sX\n' % (args_seen, STDERR))
    monkeypatch.setenv("NOHUMAN_STOCK_KRAKEN2", "1")
    argv = ["--threads", "1", "--db", "/some/db", "--output", "/dev/null", "--confidence", "0",
            "--unclassified-out", "tmp/kraken_out.fq", "reads.fq"]
    r = CommandRunner(exe)
    with caplog.at_level(logging.INFO, logger="nohuman"):
        r.run(argv)
    assert args_seen.read_text().split("\n")[:-1] == argv
    assert (r.last_stats.total_sequences, r.last_stats.classified, r.last_stats.unclassified) == (1000, 250, 750)
    assert "250 / 1000 (25.00%) sequences classified as human; 750 (75.00%) as non-human" in caplog.text


def test_stock_subprocess_failure_text(tmp_path, monkeypatch):
    exe = _fake_kraken2(tmp_path, 'echo "kraken2: database (\\"/x\\") does not contain necessary file taxo.k2d" >&2\nexit 2\n')
    monkeypatch.setenv("NOHUMAN_STOCK_KRAKEN2", "1")
    with pytest.raises(OSError) as ei:
        CommandRunner(exe).run(["--db", "/x", "r.fq"])
    assert str(ei.value).startswith(exe + " failed with stderr kraken2: database")
    with pytest.raises(OSError):  # binary missing: Command::output()? fails
        CommandRunner(str(tmp_path / "no_such_kraken2")).run([])
