"""CPU tests of the oracle: the C state-machine restatement (oracle/k2_oracle.c) against the
committed golden vectors (made by the closed-form Python restatement) and against that
restatement on fresh random inputs.  No GPU."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import k2_literal as lit
from oracle import minidb
from oracle import oracle as orc
from tests import synth
from tests.fastq_util import read_fastq

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold_db():
    return orc.OracleDB(directory=os.path.join(GOLD, "toy_db"))


def test_kat_primitives():
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    L = orc.lib()
    for x, y in kat["fmix64"]:
        assert L.k2o_fmix64(int(x)) == int(y)
    for x, n, rv, y in kat["revcomp"]:
        assert L.k2o_reverse_complement(int(x), n, rv) == int(y)
    # independent anchors: fmix64 is a bijection fixing 0; the spaced mask of SURVEY.md A.1
    assert L.k2o_fmix64(0) == 0
    assert int(kat["default_spaced_mask"]) == 0x3FFFFFFFF3333333
    # revcomp of a palindrome (ACGT) is itself; of AAAA is TTTT
    acgt = 0b00011011
    assert L.k2o_reverse_complement(acgt, 4, 1) == acgt
    assert L.k2o_reverse_complement(0, 4, 1) == 0xFF


def _check_against_golden(gold_db, fastqs, expected_json, paired, rule=1):
    if rule == 0:
        expected_json = expected_json.replace(".json", "_rule0.json")
    exp = json.load(open(os.path.join(GOLD, expected_json)))
    assert exp["meta"]["ambiguity_rule"] == rule
    gold_db.set(ambiguity_rule=rule)
    recs = [read_fastq(os.path.join(GOLD, f)) for f in fastqs]
    n = len(recs[0])
    assert n == len(exp["records"])
    frags = [tuple(r[i][2] for r in recs) if paired else recs[0][i][2] for i in range(n)]
    bases, offs = orc.pack_reads(frags, paired)
    ext = gold_db.external_ids
    assert list(ext) == exp["meta"]["external_ids"]
    for conf in exp["meta"]["confidences"]:
        out, lookups, taxa, toff = gold_db.classify(bases, offs, paired, conf, want_taxa=True)
        for i, rec in enumerate(exp["records"]):
            want = rec["by_conf"][str(conf)]
            got = [int(out[i][f]) for f in ("call", "total_kmers", "clade_hits", "hit_groups")]
            assert got == want, (i, conf, got, want)
            assert int(lookups[i]) == rec["lookups"]
            hl = _hitlist(ext, taxa[int(toff[i]):int(toff[i + 1])])
            assert hl == rec["hitlist"], (i, hl, rec["hitlist"])
    return exp


def _hitlist(ext, taxa):
    if len(taxa) == 0:
        return "0:0"
    parts, i = [], 0
    while i < len(taxa):
        j = i
        while j < len(taxa) and taxa[j] == taxa[i]:
            j += 1
        t = int(taxa[i])
        if t == orc.BORDER:
            parts += ["|:|"] * (j - i)
        elif t == orc.AMBIG:
            parts.append("A:%d" % (j - i))
        else:
            parts.append("%d:%d" % (int(ext[t]), j - i))
        i = j
    return " ".join(parts)


@pytest.mark.parametrize("rule", [1, 0])
def test_golden_single_end(gold_db, rule):
    exp = _check_against_golden(gold_db, ["reads_se.fq"], "expected_se.json", False, rule)
    calls = {r["by_conf"]["0.0"][0] for r in exp["records"]}
    assert len(calls) >= 6  # LCA calls at several depths are present in the fixture


@pytest.mark.parametrize("rule", [1, 0])
def test_golden_paired_end(gold_db, rule):
    _check_against_golden(gold_db, ["reads_pe_1.fq", "reads_pe_2.fq"], "expected_pe.json", True, rule)


def test_isolated_n_costs_k_minus_1_or_l_kmers(toy, toy_oracle):
    """The two recollections of upstream's ambiguity test: an isolated N in the middle of a
    read makes k-1 = 34 k-mers ambiguous under rule 1 (mmscanner.h is_ambiguous()), l = 31 under rule 0."""
    _, _, _, genomes, _ = toy
    g = genomes[111]
    read = g[:70] + b"N" + g[71:150]
    try:
        for rule, want in ((1, 34), (0, 31)):
            toy_oracle.set(ambiguity_rule=rule)
            _, amb = toy_oracle.scan(read)
            assert int(amb.sum()) == want
            # k-mer i ends at base i + 34: the first ambiguous one ENDS at the N, the last one is `want` further on
            assert amb[35] == 0 and amb[36] == 1 and amb[36 + want - 1] == 1 and amb[36 + want] == 0
    finally:
        toy_oracle.set(ambiguity_rule=1)


def test_lattice_reads_separate_the_switches(toy):
    """The inputs of the self-diagnosing kraken2 pin (tests/pin_lattice.py) tell every combination of the four
    unverified switches apart -- except the per-mate reset at minimum_hit_groups = 1, which no input can show."""
    import itertools
    from tests import pin_lattice
    ob, tb, hb, genomes, _ = toy
    odb = orc.OracleDB(ob, tb, hb)
    reads = pin_lattice.lattice_reads(genomes)
    bases, offs = orc.pack_reads(reads, True)
    sig = {}
    for lp, rs, ar, mh in itertools.product(*(pin_lattice.LATTICE[k] for k in
                                              ("linear_probing", "reset_per_mate", "ambiguity_rule", "minimum_hit_groups"))):
        odb.set(linear_probing=lp, reset_per_mate=rs, ambiguity_rule=ar, minimum_hit_groups=mh)
        out, _, taxa, _ = odb.classify(bases, offs, True, 0.0, want_taxa=True)
        sig.setdefault((out["call"].tobytes(), taxa.tobytes()), []).append((lp, rs, ar, mh))
    for group in sig.values():
        assert len(group) == 1 or (len(group) == 2 and all(c[3] == 1 for c in group)
                                   and group[0][0] == group[1][0] and group[0][2] == group[1][2]), group
    assert len(sig) == 20


def test_multithreaded_equals_serial(toy, toy_oracle):
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(3)
    reads = synth.sample_reads(rng, genomes, 5000, paired=True, len_jitter=50)
    bases, offs = orc.pack_reads(reads, True)
    a, la = toy_oracle.classify(bases, offs, True, 0.2)
    b, lb = toy_oracle.classify(bases, offs, True, 0.2, threads=4)
    assert np.array_equal(a, b) and np.array_equal(la, lb)


_alphabet = st.sampled_from(list(b"ACGTacgtNnRX-"))


@settings(max_examples=150, deadline=None)
@given(st.lists(_alphabet, min_size=0, max_size=260).map(bytes), st.sampled_from([0, 1]))
def test_scanner_state_machine_equals_closed_form(toy, toy_oracle, seq, rule):
    """mmscanner state machine (C) == closed form of SURVEY.md A.3 (Python), incl. ambiguity, under both rules."""
    ob, tb, hb, _, _ = toy
    ldb = _literal(ob, tb, hb)
    ldb.ambiguity_rule = rule
    toy_oracle.set(ambiguity_rule=rule)
    try:
        mins, amb = toy_oracle.scan(seq)
        want = lit.kmer_minimizers(ldb, seq)
    finally:
        ldb.ambiguity_rule = 1
        toy_oracle.set(ambiguity_rule=1)
    assert len(want) == len(mins)
    for (wa, wm), m, a in zip(want, mins, amb):
        assert bool(a) == wa
        if not wa:
            assert int(m) == wm


_LIT = {}


def _literal(ob, tb, hb):
    key = id(hb)
    if key not in _LIT:
        _LIT[key] = lit.DB.from_images(ob, tb, hb)
    return _LIT[key]


@pytest.mark.parametrize("kw", [dict(k=31, l=31), dict(k=40, l=25, spaced_mask=0),
                                dict(revcom_version=0), dict(min_hash=1 << 62), dict(k=35, l=31),
                                # k > 2l: the window reaches l-mers BEFORE an ambiguous base; the scanner
                                # drops them (queue cleared at the base), ADVICE r1
                                dict(k=19, l=8, spaced_mask=0), dict(k=35, l=15), dict(k=31, l=10)])
@pytest.mark.parametrize("rule", [1, 0])
def test_variants_c_equals_literal(kw, rule):
    ob, tb, hb, genomes, _ = synth.toy_db(seed=5, **kw)
    odb = orc.OracleDB(ob, tb, hb)
    odb.set(ambiguity_rule=rule)
    ldb = lit.DB.from_images(ob, tb, hb)
    ldb.ambiguity_rule = rule
    rng = np.random.default_rng(2)
    reads = synth.sample_reads(rng, genomes, 120, paired=False, len_jitter=80,
                               n_rate=0.02 if kw.get("k", 35) > 2 * kw.get("l", 31) else 0.002)
    bases, offs = orc.pack_reads(reads, False)
    out, lookups, taxa, toff = odb.classify(bases, offs, False, 0.15, want_taxa=True)
    for i, r in enumerate(reads):
        call, tk, ch, hg, tl, nl = lit.classify_fragment(ldb, (r,), 0.15)
        assert (int(out[i]["call"]), int(out[i]["total_kmers"]), int(out[i]["clade_hits"]),
                int(out[i]["hit_groups"])) == (call, tk, ch, hg)
        assert list(taxa[int(toff[i]):int(toff[i + 1])]) == tl
        assert nl == lookups[i]


def test_double_hashing_switch():
    ob, tb, hb, genomes, _ = synth.toy_db(seed=9, linear_probing=False)
    odb = orc.OracleDB(ob, tb, hb)
    odb.set(linear_probing=False)
    ldb = lit.DB.from_images(ob, tb, hb)
    ldb.linear_probing = False
    rng = np.random.default_rng(4)
    reads = synth.sample_reads(rng, genomes, 100, frac_random=0.1)
    bases, offs = orc.pack_reads(reads, False)
    out, _ = odb.classify(bases, offs, False, 0.0)
    for i, r in enumerate(reads):
        assert int(out[i]["call"]) == lit.classify_fragment(ldb, (r,), 0.0)[0]
    assert (out["call"] != 0).sum() > 50


def test_db_directory_shapes(tmp_path, toy):
    """validate_db_directory semantics: <dir> or <dir>/db (/root/reference/src/lib.rs:119-141)."""
    ob, tb, hb, _, _ = toy
    minidb.write_db(tmp_path / "a", ob, tb, hb)
    minidb.write_db(tmp_path / "b" / "db", ob, tb, hb)
    assert orc.OracleDB(directory=str(tmp_path / "a")).k == 35
    assert orc.OracleDB(directory=str(tmp_path / "b")).k == 35
    with pytest.raises(RuntimeError):
        orc.OracleDB(directory=str(tmp_path / "missing"))


def test_malformed_images_rejected(toy):
    ob, tb, hb, _, _ = toy
    with pytest.raises(RuntimeError):
        orc.OracleDB(ob, tb, hb[:-4])
    with pytest.raises(RuntimeError):
        orc.OracleDB(ob, b"XXXXXXXX" + tb[8:], hb)
