"""Pin-day kit: when a `kraken2` binary is on PATH, bench.py's cpu_baseline leg times exactly the argv
nohuman builds for its subprocess (/root/reference/src/main.rs:210-267) against the engine's own table written as a kraken2
database directory, and reports kind "kraken2" with the port beside it.  No kraken2 exists on either box: a fake binary
records what it was called with and answers in kraken2's stderr grammar (/root/reference/src/lib.rs:61-97)."""
import json
import os
import stat
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE = r'''#!/usr/bin/python3
import json, os, sys, time
a = sys.argv[1:]
with open(os.environ["FAKE_K2_LOG"], "a") as f:
    f.write(json.dumps(a) + "\n")
paired = "--paired" in a
out = a[a.index("--unclassified-out") + 1]
inputs = a[-2:] if paired else a[-1:]
n = sum(1 for _ in open(inputs[0], "rb")) // 4
for p in ([out.replace("#", "_1"), out.replace("#", "_2")] if paired else [out]):
    open(p, "wb").close()
for f in ("hash.k2d", "opts.k2d", "taxo.k2d"):
    assert os.path.getsize(os.path.join(a[a.index("--db") + 1], f)) > 0
time.sleep(0.05)
sys.stderr.write("Loading database information... done.\n")
sys.stderr.write("%d sequences (%.2f Mbp) processed in 0.500s (%.1f Kseq/m, %.2f Mbp/m).\n" % (n, n * 300 / 1e6, n / 0.5 * 60 / 1e3, 1.0))
sys.stderr.write("  0 sequences classified (0.00%%)\n  %d sequences unclassified (100.00%%)\n" % n)
'''


def _fake_kraken2(tmp_path):
    d = tmp_path / "bin"
    d.mkdir()
    exe = d / "kraken2"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return str(d), str(exe)


def test_the_argv_is_the_references_token_for_token():
    import bench
    # src/main.rs:215-224 (threads, db, output, confidence), :231 (--paired), :262 (--unclassified-out), :266 (inputs)
    assert bench.kraken2_argv(16, "/db", "/t/kraken_out#.fq", ["a_1.fq", "a_2.fq"]) == [
        "--threads", "16", "--db", "/db", "--output", "/dev/null", "--confidence", "0", "--paired",
        "--unclassified-out", "/t/kraken_out#.fq", "a_1.fq", "a_2.fq"]
    assert bench.kraken2_argv(1, "/db", "/t/kraken_out.fq", ["a.fq"], confidence=0.1) == [
        "--threads", "1", "--db", "/db", "--output", "/dev/null", "--confidence", "0.1", "--unclassified-out", "/t/kraken_out.fq", "a.fq"]


def test_the_stock_binary_is_timed_with_that_argv_and_its_summary_parsed(tmp_path, monkeypatch):
    import bench
    from oracle import minidb
    from tests import synth
    bindir, exe = _fake_kraken2(tmp_path)
    log = tmp_path / "argv.log"
    monkeypatch.setenv("FAKE_K2_LOG", str(log))
    ob, tb, hb, _, _ = synth.toy_db()
    cap, size, kb, vb = struct.unpack("<4Q", hb[:32])
    cells = np.frombuffer(hb[32:], dtype=np.uint32)
    db = tmp_path / "db"
    bench.write_k2_db(str(db), ob, tb, (cap, size, kb, vb), cells)
    assert (db / "hash.k2d").read_bytes() == hb and (db / "opts.k2d").read_bytes() == ob and (db / "taxo.k2d").read_bytes() == tb
    r1, r2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
    for p, t in ((r1, 1), (r2, 2)):
        p.write_bytes(b"".join(b"@syn.%d/%d\nACGT\n+\nIIII\n" % (i, t) for i in range(37)))
    res = bench.time_kraken2(exe, 8, str(db), [str(r1), str(r2)], str(tmp_path))
    assert (res["sequences"], res["classified"], res["unclassified"], res["own_seconds"]) == (37, 0, 37, 0.5)
    argv = json.loads(log.read_text().splitlines()[0])
    assert argv == bench.kraken2_argv(8, str(db), str(tmp_path / "kraken_out#.fq"), [str(r1), str(r2)])
    assert not (tmp_path / "kraken_out_1.fq").exists()  # the outputs of the timed run are removed


@pytest.mark.gpu
def test_bench_reports_kind_kraken2_when_the_binary_is_on_path(tmp_path):
    bindir, _ = _fake_kraken2(tmp_path)
    log = tmp_path / "argv.log"
    env = dict(os.environ, PATH=bindir + os.pathsep + os.environ.get("PATH", ""), FAKE_K2_LOG=str(log),
               NOHUMAN_BENCH_LOGDIR=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "20000",
                          "--capacity", "4000037", "--no-e2e", "--no-variants", "--cpu-seconds", "1"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    cb = line["cpu_baseline"]
    assert cb["kind"] == "kraken2" and cb["port"]["gpu_equals_oracle"] is True and cb["fragments"] == 20000
    assert cb["classified_equal_gpu"] is True and cb["one_thread"]["fragments"] == 2000
    calls = [json.loads(x) for x in log.read_text().splitlines()]
    assert [c[1] for c in calls] == [str(cb["cores"]), "1"]  # nproc threads, then one
    for c in calls:
        assert c[0] == "--threads" and c[2] == "--db" and c[4:9] == ["--output", "/dev/null", "--confidence", "0", "--paired"]
        assert c[9] == "--unclassified-out" and c[10].endswith("kraken_out#.fq") and len(c) == 13


@pytest.mark.gpu
def test_the_bench_line_keeps_the_drivers_contract_and_carries_its_numbers_inside_config_and_roofline(tmp_path):
    """One whole `python bench.py` at toy sizes: the contract's keys, e2e numbers inside `config`, variants inside `roofline`
    (the objects the driver keeps whole), every leg checked against its oracle / its inputs, and a line short enough to survive."""
    env = dict(os.environ, NOHUMAN_BENCH_LOGDIR=str(tmp_path), NOHUMAN_BENCH_ONT_READS="3000", NOHUMAN_BENCH_ONT_MEMBERS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--pairs", "50000", "--capacity", "400000009",
                          "--e2e-pairs", "20000", "--e2e-reps", "2", "--e2e-distinct", "2", "--cpu-seconds", "1", "--wake-ms", "20"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 12000
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None and d["dtype"] == "u64"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["value_two_streams"] > 0 and r["wake_launches"] >= 1
    assert set(r["variants"]) == {"se", "hit", "ont", "wide", "pe250", "n"}
    for name, v in r["variants"].items():
        assert v["gpu_equals_oracle"] is True and v["frac"] > 0 and v["kernel_ms"] > 0 and v["value_two_streams"] > 0, name
        # the driver's record keeps scalars only -- the same numbers as roofline.<variant>_<what>
        assert r[name + "_frac"] == v["frac"] and r[name + "_kernel_ms"] == v["kernel_ms"] and r[name + "_equals_oracle"] is True, name
        assert r[name + "_value"] == v["value"] and r[name + "_frac_two_streams"] == v["frac_two_streams"], name
    assert 0 < r["n_frac"] and r["variants"]["n"]["lookups_per_read"] < r["variants"]["se"]["lookups_per_read"]  # (N kills look-ups)
    assert r["kernel_ms_without_wake"] > 0 and r["value_without_wake"] > 0
    e = d["config"]["e2e"]
    assert e["outputs_equal_inputs"] is True and e["gzip_to_plain"]["value"] > 0 and e["gzip_to_gzip"]["value"] > 0
    assert e["ont_gzip_to_gzip"]["outputs_equal_inputs"] is True and e["ont_gzip_to_gzip"]["reads"] == 6000
    assert set(e["readers_by_name"]) == {"device", "host"} and e["gzip_encoder"]["inflates_to_the_text"] is True
    c = d["config"]
    assert c["e2e_gzip_to_plain"] == e["gzip_to_plain"]["value"] and c["e2e_gzip_to_gzip"] == e["gzip_to_gzip"]["value"]
    assert c["e2e_input_side"] == e["input_side_only"]["value"] and c["e2e_ont_gzip_to_gzip"] == e["ont_gzip_to_gzip"]["value"]
    assert c["e2e_outputs_equal_inputs"] is True and isinstance(c["e2e_reader"], str) and c["e2e_pairs"] == 40000
    assert c["e2e_host_reader_gzip_to_gzip"] == e["readers_by_name"]["host"]["gzip_to_gzip"]
    assert all(not isinstance(v, (dict, list)) for k, v in c.items() if k.startswith("e2e_"))
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["gpu_equals_oracle"] is True
    assert isinstance(d["config"]["workload"], str) and len(d["config"]["workload"]) < 200
    details = json.load(open(tmp_path / "bench_details.json"))
    assert "e2e" in details and "variants" in details and "headline_workload" in details
