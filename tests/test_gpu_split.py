"""Long reads cut into segments that different waves classify (SplitBufs in nh_device.h: a prepass lists
work items, a segment inherits kraken2's last_minimizer / last_taxon from the tile before it, partial
(taxon, count) lists are added up by the segment that finishes last).  Everything a cut could disturb is
compared with the CPU oracle, which knows no segments: runs and hit groups across a cut, ambiguous
stretches before a cut (the inherited minimizer lies further back, or does not exist), the per-k-mer taxa
list, the look-up count, many taxa per segment (second pass), buffers too small for every segment."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu

TQ = 124                 # k-mers per tile at k=35, l=31
SEG = 32 * TQ            # k-mers per segment
SPLIT_MIN = 48 * TQ      # reads of more k-mers than this are cut


def _same(got, exp, what):
    for f in ("call", "total_kmers", "clade_hits", "hit_groups"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert bad.size == 0, "%s: %s differs at %s (got %s, oracle %s)" % (what, f, bad[:5], got[f][bad[:5]], exp[f][bad[:5]])


def _genome_walk(rng, genomes, ln, sub=0.02):
    """`ln` bases stitched from random stretches of the toy genomes and random sequence"""
    allg = b"".join(genomes[k] for k in sorted(genomes))
    parts, have = [], 0
    while have < ln:
        if rng.random() < 0.6:
            st = int(rng.integers(0, len(allg) - 900))
            p = allg[st:st + int(rng.integers(150, 900))]
        else:
            p = synth.random_seq(rng, int(rng.integers(100, 1500)))
        parts.append(p)
        have += len(p)
    return synth.mutate(rng, b"".join(parts)[:ln], sub, 0.0, 0.02)


def _check(eng, odb, reads, conf=0.0, what=""):
    bases, offs = orc.pack_reads(reads, False)
    exp, lookups, etaxa, etoff = odb.classify(bases, offs, False, conf, want_taxa=True)
    eng.reset_stats()
    got, taxa, toff = eng.classify(bases, offs, False, conf, want_taxa=True, long_reads=True)
    _same(got, exp, what)
    assert np.array_equal(toff, etoff) and np.array_equal(taxa, etaxa), what
    st = eng.stats()
    assert st.total_sequences == len(reads) and st.total_bases == bases.size
    assert st.table_lookups == int(lookups.sum()), what  # the inherited minimizer's look-up is not one kraken2 makes
    assert st.classified == int((exp["call"] != 0).sum())
    return exp


def test_lengths_around_the_cut_rules(toy, toy_oracle, toy_engine):
    """k-mer counts around every rule: the split threshold, whole segments, one k-mer more or less."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(21)
    nks = [SPLIT_MIN - 1, SPLIT_MIN, SPLIT_MIN + 1, SPLIT_MIN + TQ, 2 * SEG - 1, 2 * SEG, 2 * SEG + 1, 3 * SEG + 5,
           3 * SEG - TQ, 3 * SEG - TQ + 1, 7 * SEG + 17, 35, 1, 0]
    reads = [_genome_walk(rng, genomes, nk + 34) if nk else b"ACGT" for nk in nks]
    reads += [synth.random_seq(rng, 150), b"", _genome_walk(rng, genomes, 40_000)]
    for conf in (0.0, 0.2):
        exp = _check(toy_engine, toy_oracle, reads, conf, "lengths conf=%s" % conf)
    assert (exp["call"] != 0).sum() >= 10 and exp["hit_groups"].max() > 50


def test_ambiguous_stretches_before_a_cut(toy, toy_oracle, toy_engine):
    """The minimizer a segment inherits is that of the last UNAMBIGUOUS k-mer before it: Ns of every length
    right before the cut (less than a tile, several tiles, everything back to the start of the read), Ns
    straddling the cut, Ns right behind it."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(22)
    reads = []
    for cut in (SEG, 2 * SEG):
        for n_before, n_after in ((1, 0), (30, 0), (31, 0), (35, 0), (TQ - 1, 0), (TQ, 0), (TQ + 40, 0), (5 * TQ + 3, 0),
                                  (cut + 34, 0), (20, 20), (0, 1), (0, 70), (200, 200), (cut + 34 - 100, 0)):
            r = bytearray(_genome_walk(rng, genomes, 3 * SEG + 500))
            pos = cut + 34  # first base behind the last base of k-mer cut-1 ... the cut lies between k-mers cut-1 and cut
            lo = max(0, pos - n_before)
            r[lo:pos + n_after] = b"N" * (pos + n_after - lo)
            reads.append(bytes(r))
    # a whole read of Ns, and one with an unambiguous island only
    reads.append(b"N" * (3 * SEG))
    r = bytearray(b"N" * (3 * SEG + 100))
    r[SEG - 300:SEG + 34] = _genome_walk(rng, genomes, 334)
    reads.append(bytes(r))
    exp = _check(toy_engine, toy_oracle, reads, 0.0, "N before cut")
    assert (exp["call"] != 0).sum() >= 10


def test_one_run_across_many_cuts(toy, toy_oracle, toy_engine):
    """A low-complexity read keeps ONE minimizer for thousands of k-mers: the run starts in the first segment
    and every later segment inherits it (one hit group, or none, for the whole read)."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(23)
    g = genomes[111]
    reads = [b"A" * (4 * SEG + 200), b"AC" * (2 * SEG + 77), (g[:60] + b"T" * (3 * SEG)), b"ACGTTGCA" * 2000 + g[:300],
             g[:380] * 40]
    exp = _check(toy_engine, toy_oracle, reads, 0.0, "long runs")
    assert exp["hit_groups"][0] <= 1


def test_random_long_reads_with_every_feature(toy, toy_oracle, toy_engine, monkeypatch):
    """Lognormal lengths up to 60 kb, hits, 0.3 % N, lower case; the same batch with the buffers too small for
    most segments (those reads go whole), with cutting switched off, and in ten pieces."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(24)
    reads = []
    for _ in range(260):
        ln = int(np.clip(rng.lognormal(8.6, 0.9), 100, 60_000))
        reads.append(synth.mutate(rng, _genome_walk(rng, genomes, ln), 0.0, 0.003, 0.0))
    exp = _check(toy_engine, toy_oracle, reads, 0.05, "random long")
    assert sum(len(r) - 34 > SPLIT_MIN for r in reads) > 60
    bases, offs = orc.pack_reads(reads, False)
    from nohuman_amd import Engine
    ob, tb, hb, _, _ = toy
    for env in ({"NOHUMAN_SEG_CAP": "40"}, {"NOHUMAN_SEG_CAP": "1"}, {"NOHUMAN_NO_SPLIT": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if "NOHUMAN_NO_SPLIT" in env:
            # the switch is read once per process: a child interpreter takes the measurement
            import subprocess, sys, os
            code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
                    "from tests import synth; from oracle import oracle as orc; from nohuman_amd import Engine\n"
                    "ob, tb, hb, g, _ = synth.toy_db()\n"
                    "d = np.load(sys.argv[1])\n"
                    "with Engine.from_images(ob, tb, hb) as e:\n"
                    "    got = e.classify(d['bases'], d['offs'], False, 0.05, long_reads=True)\n"
                    "np.save(sys.argv[2], got)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                np.savez(os.path.join(td, "in.npz"), bases=bases, offs=offs)
                subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npy")])
                got = np.load(os.path.join(td, "out.npy"))
        else:
            with Engine.from_images(ob, tb, hb) as eng:
                got = eng.classify(bases, offs, False, 0.05, long_reads=True)
        _same(got, exp, str(env))
        for k in env:
            monkeypatch.delenv(k)
    cuts = np.linspace(0, len(reads), 11).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        o = offs[a:b + 1]
        got = toy_engine.classify(bases[int(o[0]):int(o[-1])], o - o[0], False, 0.05, long_reads=True)
        _same(got, exp[a:b], "piece %d" % a)


def test_many_taxa_in_one_segment_take_the_second_pass(monkeypatch):
    """A segment keeps at most 30 (taxon, count) pairs in its partial; more than that leaves the read to the
    second pass (whole read, one wave, 2048-entry list), like an overflow of the 64-entry list does."""
    from nohuman_amd import Engine
    from oracle import minidb
    rng = np.random.default_rng(25)
    edges = {1: 0}
    for g in range(10):
        edges[100 + g] = 1
    leaves = []
    for i in range(150):
        edges[1000 + i] = 100 + i % 10
        leaves.append(1000 + i)
    tax = minidb.Taxonomy(edges)
    segs = {e: synth.random_seq(rng, 120) for e in leaves}
    hashb, _ = minidb.build_hash(tax, sorted(segs.items()), 40009)
    odb = orc.OracleDB(minidb.opts_bytes(), tax.to_bytes(), hashb)
    reads = []
    for i in range(120):
        kind = i % 4
        if kind == 0:    # ~100 leaves: > 30 per segment, > 64 per read
            pick = rng.choice(leaves, size=int(rng.integers(90, 141)), replace=False)
            reads.append(b"".join(segs[int(e)][10:110] for e in pick))
        elif kind == 1:  # 40 leaves packed into the first segment of a long read, random sequence behind
            pick = rng.choice(leaves, size=40, replace=False)
            reads.append(b"".join(segs[int(e)][10:100] for e in pick) + synth.random_seq(rng, 9000))
        elif kind == 2:  # 20 leaves per segment, 60 in the read: every partial fits, the sum does too
            pick = rng.choice(leaves, size=60, replace=False)
            reads.append(b"".join(segs[int(e)][10:110] + synth.random_seq(rng, 100) for e in pick))
        else:
            reads.append(synth.random_seq(rng, int(rng.integers(150, 9000))))
    with Engine.from_images(minidb.opts_bytes(), tax.to_bytes(), hashb) as eng:
        exp = _check(eng, odb, reads, 0.0, "many taxa")
    assert (exp["call"] != 0).sum() > 60


@pytest.mark.parametrize("seed", range(8))
def test_segments_soak_many_reads_many_launches(toy, toy_oracle, toy_engine, seed):
    """the partials of a split read travel by write-through stores, a counter and sc1 loads (no release /
    acquire pair) -- a soak over seeds and repeated launches: reads of 20-200 kb (up to fifty segments each, finished by
    whichever wave comes last), every launch compared with the oracle, which knows no segments."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(500 + seed)
    reads = []
    for _ in range(40):
        ln = int(rng.integers(20_000, 200_000))
        reads.append(synth.mutate(rng, _genome_walk(rng, genomes, ln), 0.0, 0.002, 0.0))
    bases, offs = orc.pack_reads(reads, False)
    exp = toy_oracle.classify(bases, offs, False, 0.0)[0]
    for launch in range(4):
        got = toy_engine.classify(bases, offs, False, 0.0, long_reads=True)
        _same(got, exp, "seed %d launch %d" % (seed, launch))
