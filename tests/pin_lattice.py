"""Self-diagnosing half of the kraken2 pin.  TEST INFRASTRUCTURE.

Every kraken2 rule this build could not verify against a binary is a runtime switch of the engine
(`nh_options`, include/nohuman_engine.h):

    linear_probing      1 | 0          compact_hash.cc built with -DLINEAR_PROBING or with double hashing
    reset_per_mate      1 | 0          last_minimizer / last_taxon reset for each mate (A.5)
    ambiguity_rule      1 | 0          mmscanner.h is_ambiguous() | the bool* flag of NextMinimizer (A.3 (i)/(ii))
    minimum_hit_groups  2 | 1 | 3      default of the kraken2 wrapper's --minimum-hit-groups (A.5); 0 cannot be told
                                       from 1 by any input (a call needs a hit, a hit is a hit group)

Given the per-read output of the REAL program (`kraken2 --output`, the reference's argv
/root/reference/src/main.rs:215-267) and the same inputs, `diagnose()` runs the engine over the whole lattice
(2 x 2 x 2 x 3 = 24 runs of a small input) and names the combination(s) that reproduce kraken2's lines, or the
closest ones with their first differing read -- so ONE run on any box that has the binary pins every switch.

    python -m tests.pin_lattice <db_dir> <kraken2_output.txt> <confidence> <reads_1.fq> [reads_2.fq]
"""
from __future__ import annotations

import itertools
import os
import sys
import tempfile

LATTICE = {
    "linear_probing": (1, 0),
    "reset_per_mate": (1, 0),
    "ambiguity_rule": (1, 0),
    "minimum_hit_groups": (2, 1, 3),
}


def lattice_reads(genomes, n_plain=300, n_crafted=600, seed=5):
    """Read pairs on which the switches SHOW: reads drawn from the toy genomes (N at 0.4 %: the ambiguity rule; the
    table's collision runs: the probing rule) plus pairs crafted around the hit-group threshold -- 0..3 short genome
    pieces in random sequence (1, 2, 3 hit groups), and every other pair has the SAME 35-mer at the end of mate 1 and
    at the start of mate 2 (one hit group more iff the scanner's last minimizer is reset per mate).  On these,
    every combination of the lattice gives different kraken lines except the per-mate reset at minimum_hit_groups
    = 1, which no input can show (tests/test_oracle.py::test_lattice_reads_separate_the_switches).  Run at confidence 0."""
    import numpy as np
    from tests import synth
    rng = np.random.default_rng(seed)
    reads = synth.sample_reads(rng, genomes, n_plain, paired=True, len_jitter=40, n_rate=0.004)
    leaf = sorted(genomes)[2] if 111 not in genomes else 111
    g = genomes[leaf]
    own = g.rfind(b"N") + 1  # the leaf's own segment (its ancestors' shared segments come first)
    lo, hi = own + 7, len(g) - 80
    for i in range(n_crafted):
        a = synth.random_seq(rng, 30)
        for _ in range(int(rng.integers(0, 4))):
            y = int(rng.integers(lo, hi))
            a += synth.random_seq(rng, int(rng.integers(20, 40))) + g[y:y + int(rng.integers(35, 38))]
        x = int(rng.integers(lo, hi))
        share = i % 2 == 0
        m1 = a + (g[x:x + 35] if share else synth.random_seq(rng, 20))
        m2 = (g[x:x + 35] if share else b"") + synth.random_seq(rng, 100)
        reads.append((m1, m2))
    return reads


def combos():
    keys = list(LATTICE)
    for vals in itertools.product(*(LATTICE[k] for k in keys)):
        yield dict(zip(keys, vals))


def engine_lines(eng, inputs, confidence, opts, workdir):
    """kraken-output lines of the (open) engine under the given switches (nh_run_engine through the C ABI)."""
    paired = len(inputs) == 2
    tag = "_".join("%s%d" % (k[0], v) for k, v in opts.items())
    kout = os.path.join(workdir, "k_%s.txt" % tag)
    eng.set_options(**opts)
    eng.run(inputs[0], os.path.join(workdir, "o1.fq"), in2=inputs[1] if paired else None,
            out2=os.path.join(workdir, "o2.fq") if paired else None, kraken_output=kout, confidence=float(confidence))
    with open(kout) as f:
        lines = f.read().splitlines()
    os.unlink(kout)
    return lines


def diagnose(db_dir, ref_lines, inputs, confidence, workdir=None, out=sys.stdout):
    """Returns the list of switch combinations whose engine output equals ref_lines (empty: none does)."""
    own = workdir is None
    if own:
        tmp = tempfile.TemporaryDirectory()
        workdir = tmp.name
    from nohuman_amd import Engine
    rows = []
    eng = Engine.open(db_dir)
    keep = eng.options()
    for opts in combos():
        lines = engine_lines(eng, inputs, confidence, opts, workdir)
        bad = [i for i, (a, b) in enumerate(zip(ref_lines, lines)) if a != b]
        nbad = len(bad) + abs(len(ref_lines) - len(lines))
        rows.append((nbad, opts, bad[0] if bad else None, lines))
    eng.close()
    del keep
    rows.sort(key=lambda r: r[0])
    exact = [r[1] for r in rows if r[0] == 0]
    print("kraken2 pin, switch lattice over %d reads (differing lines per combination):" % len(ref_lines), file=out)
    for nbad, opts, first, _ in rows:
        print("  %6d  %s" % (nbad, " ".join("%s=%d" % kv for kv in opts.items())), file=out)
    if exact:
        print("REPRODUCES kraken2: " + " | ".join(" ".join("%s=%d" % kv for kv in o.items()) for o in exact), file=out)
        default = {"linear_probing": 1, "reset_per_mate": 1, "ambiguity_rule": 1, "minimum_hit_groups": 2}
        if default in exact:
            print("the engine's DEFAULTS are among them: every switch is pinned as shipped", file=out)
        else:
            print("the engine's defaults are NOT among them: change the defaults in nh_engine.hip / k2_oracle.c "
                  "(include/nohuman_engine.h NH_AMBIGUITY_DEFAULT) to the combination above and record it in BASELINE.md section 2",
                  file=out)
    else:
        nbad, opts, first, lines = rows[0]
        print("NO combination reproduces kraken2; closest: %s (%d lines differ)" % (opts, nbad), file=out)
        if first is not None:
            print("  first differing read %d:\n    kraken2: %s\n    engine : %s" % (first, ref_lines[first], lines[first]), file=out)
        print("  (reads whose lines differ under EVERY combination point at a rule outside the lattice)", file=out)
        print("  -> a rule outside the lattice differs (formats, hit-list layout, record parsing): see SURVEY.md Appendix A", file=out)
    if own:
        tmp.cleanup()
    return exact


def main(argv):
    if len(argv) < 4:
        print(__doc__)
        return 2
    db_dir, ref, conf, *inputs = argv
    ref_lines = open(ref).read().splitlines()
    exact = diagnose(db_dir, ref_lines, inputs, conf)
    return 0 if exact else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
