"""Parity pin against the REAL kraken2.  Runs by itself wherever `kraken2` and
`kraken2-build` are on PATH (they are neither in the build container nor on the GPU boxes of this
pool: the test then SKIPS and says so, and every status table keeps "parity vs kraken2: UNPINNED").

What it does when the tools exist -- no network needed:
  1. builds a small custom database with the real builder from the toy genomes (hand-written
     names.dmp / nodes.dmp, `kraken:taxid|N` FASTA headers, --no-masking), which also pins the
     on-disk formats: the engine must load what kraken2-build wrote;
  2. runs the exact argv nohuman builds (/root/reference/src/main.rs:215-267) once through the stock
     binary (CommandRunner with NOHUMAN_STOCK_KRAKEN2=1, /root/reference/src/lib.rs:22-48) and once
     through the engine, single-end and paired, two confidence values;
  3. compares the per-read kraken output lines (C/U, id, taxid, lengths, hit list), the kept-read
     FASTQ bytes and the three summary integers.
A database can also be handed in: NOHUMAN_PIN_DB=<dir with hash/opts/taxo.k2d> (e.g. HPRC.r2) and
NOHUMAN_PIN_READS=<fastq>[,<fastq mate 2>]."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import synth
from tests.fastq_util import write_fastq

pytestmark = pytest.mark.gpu


def _need(tool):
    path = shutil.which(tool)
    if path is None:
        reason = "%s is not on PATH: parity vs kraken2 stays UNPINNED on this box" % tool
        print("SKIP:", reason)
        pytest.skip(reason)
    return path


def _build_real_db(tmp_path, genomes):
    _need("kraken2-build")
    db = tmp_path / "realdb"
    (db / "taxonomy").mkdir(parents=True)
    ranks = {1: "no rank", 10: "phylum", 20: "phylum", 11: "genus", 12: "species", 111: "species", 112: "species",
             21: "species", 9606: "species"}
    with open(db / "taxonomy" / "nodes.dmp", "w") as f:
        for tid, parent in synth.TOY_EDGES.items():
            f.write("%d\t|\t%d\t|\t%s\t|\t-\t|\n" % (tid, parent if parent else 1, ranks[tid]))
    with open(db / "taxonomy" / "names.dmp", "w") as f:
        for tid in synth.TOY_EDGES:
            f.write("%d\t|\ttaxon%d\t|\t\t|\tscientific name\t|\n" % (tid, tid))
    fa = tmp_path / "lib.fa"
    with open(fa, "w") as f:
        for tid, seq in sorted(genomes.items()):
            f.write(">g%d|kraken:taxid|%d\n%s\n" % (tid, tid, seq.decode()))
    for cmd in (["kraken2-build", "--add-to-library", str(fa), "--db", str(db), "--no-masking"],
                ["kraken2-build", "--build", "--db", str(db), "--threads", "2", "--no-masking"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("kraken2-build failed here (%s): %s" % (" ".join(cmd[:2]), r.stderr[-400:]))
    return str(db)


def _inputs(tmp_path, genomes, paired):
    env_reads = os.environ.get("NOHUMAN_PIN_READS")
    if env_reads:
        parts = env_reads.split(",")
        if paired != (len(parts) == 2):
            pytest.skip("NOHUMAN_PIN_READS does not have this shape")
        return parts
    rng = np.random.default_rng(99)
    reads = synth.sample_reads(rng, genomes, 3000, paired=paired, len_jitter=40, n_rate=0.003)
    if paired:  # plus the pairs on which every switch of the lattice shows (tests/pin_lattice.py)
        from tests import pin_lattice
        reads += pin_lattice.lattice_reads(genomes)
    if paired:
        p1, p2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
        write_fastq(p1, [("pin.%d/1" % i, a) for i, (a, b) in enumerate(reads)])
        write_fastq(p2, [("pin.%d/2" % i, b) for i, (a, b) in enumerate(reads)])
        return [p1, p2]
    p = str(tmp_path / "r.fq")
    write_fastq(p, [("pin.%d" % i, a) for i, a in enumerate(reads)])
    return [p]


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("confidence", ["0", "0.1"])
def test_engine_equals_real_kraken2(tmp_path, monkeypatch, toy, paired, confidence):
    from nohuman_amd import CommandRunner
    kraken2 = _need("kraken2")
    _, _, _, genomes, _ = toy
    db = os.environ.get("NOHUMAN_PIN_DB") or _build_real_db(tmp_path, genomes)
    inputs = _inputs(tmp_path, genomes, paired)
    outs = {}
    for who in ("stock", "engine"):
        d = tmp_path / who
        d.mkdir()
        out = str(d / ("kraken_out#.fq" if paired else "kraken_out.fq"))
        argv = ["--threads", "2", "--db", db, "--output", str(d / "k.txt"), "--confidence", confidence,
                "--report", str(d / "report.txt")]
        if paired:
            argv.append("--paired")
        argv += ["--unclassified-out", out] + inputs
        if who == "stock":
            monkeypatch.setenv("NOHUMAN_STOCK_KRAKEN2", "1")
        else:
            monkeypatch.delenv("NOHUMAN_STOCK_KRAKEN2", raising=False)
        r = CommandRunner(kraken2 if who == "stock" else "kraken2")
        r.run(argv)
        outs[who] = (d, r.last_stats)
    ds, ss = outs["stock"]
    de, se = outs["engine"]
    assert (ss.total_sequences, ss.classified, ss.unclassified) == (se.total_sequences, se.classified, se.unclassified)
    ref_lines = open(ds / "k.txt").read().splitlines()
    eng_lines = open(de / "k.txt").read().splitlines()
    if ref_lines != eng_lines:
        # self-diagnosing: walk the switch lattice -- probing x per-mate reset x ambiguity rule x hit
        # groups -- and print the ONE combination that reproduces kraken2's lines, so this single run pins everything
        from tests import pin_lattice
        lat = tmp_path / "lattice"
        lat.mkdir()
        exact = pin_lattice.diagnose(db, ref_lines, inputs, confidence, workdir=str(lat))
        pytest.fail("the engine's defaults do not reproduce kraken2; combinations that do: %s (table above)" % (exact or "none"))
    assert len(ref_lines) == len(eng_lines)
    for i, (a, b) in enumerate(zip(ref_lines, eng_lines)):
        assert a == b, "read %d: kraken2 says %r, the engine %r" % (i, a, b)
    names = ["kraken_out_1.fq", "kraken_out_2.fq"] if paired else ["kraken_out.fq"]
    for n in names:
        assert open(ds / n, "rb").read() == open(de / n, "rb").read(), n
    assert open(ds / "report.txt").read() == open(de / "report.txt").read()
    print("PINNED: engine == kraken2 on %d fragments (paired=%s, confidence=%s)" % (len(ref_lines), paired, confidence))


def test_the_lattice_walker_names_the_combination(tmp_path, toy):
    """No kraken2 needed: lines made by the engine under a NON-default combination play the binary's part; the walker
    must name exactly that combination (and say that the defaults are not it)."""
    import io
    from nohuman_amd import Engine
    from oracle import minidb
    from tests import pin_lattice
    ob, tb, hb, genomes, _ = toy
    db = tmp_path / "db"
    minidb.write_db(str(db), ob, tb, hb)
    reads = pin_lattice.lattice_reads(genomes)
    p1, p2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    write_fastq(p1, [("lat.%d/1" % i, a) for i, (a, b) in enumerate(reads)])
    write_fastq(p2, [("lat.%d/2" % i, b) for i, (a, b) in enumerate(reads)])
    for secret in (dict(linear_probing=1, reset_per_mate=0, ambiguity_rule=0, minimum_hit_groups=3),
                   dict(linear_probing=0, reset_per_mate=1, ambiguity_rule=1, minimum_hit_groups=2)):
        with Engine.open(str(db)) as eng:
            ref = pin_lattice.engine_lines(eng, [p1, p2], "0", secret, str(tmp_path))
        log = io.StringIO()
        exact = pin_lattice.diagnose(str(db), ref, [p1, p2], "0", out=log)
        print(log.getvalue())
        assert exact == [secret], exact
        assert "NOT among them" in log.getvalue()
    # the defaults reproduce themselves, and the walker says so
    with Engine.open(str(db)) as eng:
        ref = pin_lattice.engine_lines(eng, [p1, p2], "0", dict(linear_probing=1, reset_per_mate=1, ambiguity_rule=1,
                                                               minimum_hit_groups=2), str(tmp_path))
    log = io.StringIO()
    exact = pin_lattice.diagnose(str(db), ref, [p1, p2], "0", out=log)
    assert len(exact) == 1 and "pinned as shipped" in log.getvalue()
