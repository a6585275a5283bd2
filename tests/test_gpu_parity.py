"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu


def _assert_same(got, exp, what=""):
    for f in ("call", "total_kmers", "clade_hits", "hit_groups"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert bad.size == 0, "%s: field %s differs at %s (got %s, oracle %s)" % (
            what, f, bad[:5], got[f][bad[:5]], exp[f][bad[:5]])


def test_probe_reports_gfx950():
    import nohuman_amd
    msg = nohuman_amd.probe()
    assert "gfx950" in msg


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("confidence", [0.0, 0.1, 0.5, 1.0])
def test_toy_db_parity(toy, toy_oracle, toy_engine, paired, confidence):
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(11 + int(paired))
    reads = synth.sample_reads(rng, genomes, 2000, paired=paired, len_jitter=120)
    bases, offs = orc.pack_reads(reads, paired)
    exp, lookups, etaxa, etoff = toy_oracle.classify(bases, offs, paired, confidence, want_taxa=True)
    toy_engine.reset_stats()
    got, taxa, toff = toy_engine.classify(bases, offs, paired, confidence, want_taxa=True)
    _assert_same(got, exp, "toy paired=%s conf=%s" % (paired, confidence))
    assert np.array_equal(toff, etoff)
    assert np.array_equal(taxa, etaxa)
    st = toy_engine.stats()
    assert st.total_sequences == len(reads)
    assert st.classified == int((exp["call"] != 0).sum())
    assert st.table_lookups == int(lookups.sum())
    assert st.total_bases == bases.size
    assert (exp["call"] != 0).sum() > 30  # the hit / LCA paths are really exercised


@pytest.fixture(params=[1, 0], ids=["ambig_queue", "ambig_last_lmer"])
def both_rules(request, toy_oracle, toy_engine):
    """nh_options.ambiguity_rule on the shared engine and oracle: 1 = mmscanner.h is_ambiguous() (default), 0 = last l bases"""
    keep = toy_engine.ambiguity_rule()
    toy_oracle.set(ambiguity_rule=request.param)
    toy_engine.set_options(ambiguity_rule=request.param)
    yield request.param
    toy_oracle.set(ambiguity_rule=keep)
    toy_engine.set_options(ambiguity_rule=keep)


def test_isolated_n_costs_k_minus_1_or_l_kmers(toy, toy_engine, both_rules):
    """the hit list of a read with one N: A:34 under rule 1, A:31 under rule 0"""
    _, _, _, genomes, _ = toy
    g = genomes[111]
    read = g[:70] + b"N" + g[71:150]
    bases, offs = orc.pack_reads([read], False)
    _, taxa, _ = toy_engine.classify(bases, offs, False, 0.0, want_taxa=True)
    assert int((taxa == orc.AMBIG).sum()) == (34 if both_rules == 1 else 31)
    assert taxa[35] != orc.AMBIG and taxa[36] == orc.AMBIG


def test_ambiguity_rule_is_validated(toy_engine):
    with pytest.raises(Exception):
        toy_engine.set_options(ambiguity_rule=2)
    assert toy_engine.ambiguity_rule() in (0, 1) and toy_engine.options().ambiguity_rule in (1, 2)


def test_edge_cases(toy, toy_oracle, toy_engine, both_rules):
    """empty reads, reads shorter than l / k, all-N, N at every offset, lower case, long reads."""
    _, _, _, genomes, _ = toy
    g = genomes[111]
    reads = [b"", b"A", g[:30], g[:31], g[:34], g[:35], g[:36], b"N" * 150, g[:150].lower(),
             g[100:100 + 124 + 34], g[100:100 + 125 + 34], g[100:100 + 126 + 34],
             g[:1200], g[37:1800] + g[5:900]]
    for i in range(0, 150, 7):
        r = bytearray(g[200:350])
        r[i] = ord("N")
        reads.append(bytes(r))
    for i in (0, 1, 33, 34, 35, 36, 120, 148, 149):
        r = bytearray(g[300:450])
        r[i] = ord("n")
        r[(i * 3) % 150] = ord("R")
        reads.append(bytes(r))
    bases, offs = orc.pack_reads(reads, False)
    for conf in (0.0, 0.3):
        exp, _, etaxa, etoff = toy_oracle.classify(bases, offs, False, conf, want_taxa=True)
        got, taxa, toff = toy_engine.classify(bases, offs, False, conf, want_taxa=True)
        _assert_same(got, exp, "edge")
        assert np.array_equal(taxa, etaxa)
    # the same reads as mates of pairs (mixed lengths inside a pair)
    pairs = [(reads[i], reads[-1 - i]) for i in range(len(reads) // 2)]
    bases, offs = orc.pack_reads(pairs, True)
    exp, _, etaxa, etoff = toy_oracle.classify(bases, offs, True, 0.05, want_taxa=True)
    got, taxa, toff = toy_engine.classify(bases, offs, True, 0.05, want_taxa=True)
    _assert_same(got, exp, "edge pairs")
    assert np.array_equal(taxa, etaxa)


def test_empty_batch(toy_engine):
    out = toy_engine.classify(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert out.size == 0


def test_long_reads_parity(toy, toy_oracle, toy_engine):
    """variable-length long reads (config 4 shape, scaled down): many tiles per read."""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(5)
    allg = b"".join(genomes[k] for k in sorted(genomes))
    reads = []
    for _ in range(300):
        ln = int(np.clip(rng.lognormal(7.5, 0.8), 200, 12000))
        parts = []
        while sum(map(len, parts)) < ln:
            st = int(rng.integers(0, len(allg) - 300))
            parts.append(allg[st:st + int(rng.integers(100, 700))])
        reads.append(synth.mutate(rng, b"".join(parts)[:ln], 0.05, 0.001, 0.0))
    bases, offs = orc.pack_reads(reads, False)
    exp, _, etaxa, _ = toy_oracle.classify(bases, offs, False, 0.0, want_taxa=True)
    got, taxa, _ = toy_engine.classify(bases, offs, False, 0.0, want_taxa=True)
    _assert_same(got, exp, "long")
    assert np.array_equal(taxa, etaxa)


@pytest.mark.parametrize("variant", ["k31l31", "k40l25_nospace", "revcom0", "minhash", "double_hash"])
def test_db_variants(variant):
    """other k/l, no spaced seed, legacy revcom_version 0, min-hash subsampling, double hashing."""
    from nohuman_amd import Engine
    kw = {"k31l31": dict(k=31, l=31), "k40l25_nospace": dict(k=40, l=25, spaced_mask=0),
          "revcom0": dict(revcom_version=0), "minhash": dict(min_hash=1 << 62),
          "double_hash": dict(linear_probing=False)}[variant]
    ob, tb, hb, genomes, _ = synth.toy_db(seed=3, **kw)
    odb = orc.OracleDB(ob, tb, hb)
    rng = np.random.default_rng(8)
    reads = synth.sample_reads(rng, genomes, 600, paired=False, len_jitter=60)
    bases, offs = orc.pack_reads(reads, False)
    with Engine.from_images(ob, tb, hb) as eng:
        if variant == "double_hash":
            odb.set(linear_probing=False)
            eng.set_options(linear_probing=False)
        exp, _, etaxa, _ = odb.classify(bases, offs, False, 0.1, want_taxa=True)
        got, taxa, _ = eng.classify(bases, offs, False, 0.1, want_taxa=True)
    _assert_same(got, exp, variant)
    assert np.array_equal(taxa, etaxa)
    assert (exp["call"] != 0).sum() > 50


def test_more_than_64_distinct_taxa_take_the_second_pass():
    """A fragment that hits more than 64 distinct taxa overflows the hot kernel's LDS list and is
    finished by the BIG kernel variant (2048 entries); results stay bit-exact."""
    from nohuman_amd import Engine
    from oracle import minidb
    rng = np.random.default_rng(21)
    n_leaf = 150
    edges = {1: 0}
    for g in range(10):          # 10 genera under the root, 15 leaves each
        edges[100 + g] = 1
    leaves = []
    for i in range(n_leaf):
        ext = 1000 + i
        edges[ext] = 100 + i % 10
        leaves.append(ext)
    tax = minidb.Taxonomy(edges)
    segs = {ext: synth.random_seq(rng, 120) for ext in leaves}
    hashb, size = minidb.build_hash(tax, sorted(segs.items()), 40009)
    ob, tb = minidb.opts_bytes(), tax.to_bytes()
    assert tax.node_count > 64
    reads = []
    for _ in range(40):          # long reads visiting 70-140 different leaves
        k = int(rng.integers(70, 141))
        pick = rng.choice(leaves, size=k, replace=False)
        reads.append(b"".join(segs[int(e)][10:110] for e in pick))
    reads += synth.sample_reads(rng, {e: segs[e] for e in leaves[:20]}, 200, length=100, frac_random=0.2)
    bases, offs = orc.pack_reads(reads, False)
    odb = orc.OracleDB(ob, tb, hashb)
    for conf in (0.0, 0.02, 0.3):
        exp, lookups, etaxa, _ = odb.classify(bases, offs, False, conf, want_taxa=True)
        with Engine.from_images(ob, tb, hashb) as eng:
            got, taxa, _ = eng.classify(bases, offs, False, conf, want_taxa=True)
            st = eng.stats()
        _assert_same(got, exp, "many taxa conf=%s" % conf)
        assert np.array_equal(taxa, etaxa)
        assert st.classified == int((exp["call"] != 0).sum())
        assert st.table_lookups == int(lookups.sum())
    # the long reads really exceed the 64-entry list
    from oracle import k2_literal as lit
    ldb = lit.DB.from_images(ob, tb, hashb)
    distinct = {t for t in lit.classify_fragment(ldb, (reads[0],), 0.0)[4] if t not in (0, lit.AMBIG)}
    assert len(distinct) > 64


def test_wide_position_variant_on_small_tables():
    """Tables of 2^32 - 256 cells or more are probed with 64-bit cell positions by a separate kernel
    variant.  NOHUMAN_FORCE_WIDE routes the small test tables through that variant: the parity tests
    of this file must hold there as well (the variant is picked once per process, hence a child)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NOHUMAN_FORCE_WIDE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x",
                        "-k", "toy_db_parity or edge_cases or long_reads or distinct_taxa or k31l31"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_parity_suites_under_the_other_ambiguity_rule():
    """Every oracle-differential test of the parity, split-read and whole-run files once more with BOTH sides switched to
    ambiguity rule 0 (last l bases; the default is rule 1, mmscanner.h is_ambiguous()): the engine through
    NOHUMAN_OPT_AMBIGUITY_RULE, the oracle through K2O_AMBIGUITY_RULE, the goldens through their _rule0 files."""
    import subprocess
    import sys
    if os.environ.get("NOHUMAN_OPT_AMBIGUITY_RULE") is not None:
        pytest.skip("already the child run")
    env = dict(os.environ, NOHUMAN_OPT_AMBIGUITY_RULE="0", K2O_AMBIGUITY_RULE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_split.py",
                        "tests/test_gpu_run.py", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "not 2_pow_32 and not wide_position and not other_ambiguity and not both_rules "
                        "and not isolated_n and not allreduce and not sched"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_table_larger_than_2_pow_32_cells():
    """A 4.4 G-cell table (17.6 GB in HBM): cell positions beyond 2^32 are really addressed.  Reads
    cut from sequences whose minimizers were inserted must all classify to the inserted taxon, random
    reads must not, and the lookup count must equal the one a small table gives for the same reads."""
    import torch
    from nohuman_amd import Engine
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    n_src, L = 4000, 1000
    src = acgt[torch.randint(0, 4, (n_src * L + 64,), generator=g, device=dev)].contiguous()
    src_off = (torch.arange(n_src + 1, dtype=torch.int64, device=dev) * L).contiguous()
    n_reads, R = 20000, 150
    starts = torch.randint(0, n_src * L - R, (n_reads // 2,), generator=g, device=dev)
    starts = starts - torch.clamp((starts % L) + R - L, min=0)  # stay inside one source sequence
    idx = starts[:, None] + torch.arange(R, device=dev)[None, :]
    hit_reads = src[idx.reshape(-1)]
    rnd_reads = acgt[torch.randint(0, 4, (n_reads // 2 * R,), generator=g, device=dev)]
    bases = torch.cat([hit_reads, rnd_reads, torch.full((64,), 65, dtype=torch.uint8, device=dev)]).contiguous()
    offs = (torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * R).contiguous()
    res = {}
    for name, cap in (("small", 50_000_017), ("wide", 4_400_000_011)):
        with Engine.synthetic(cap, int(cap * 0.5), depth=30, seed=5) as eng:
            assert (eng.info.capacity >= 2 ** 32) == (name == "wide")
            eng.add_sequences(src.data_ptr(), src_off.data_ptr(), n_src, 30)
            out = torch.zeros(n_reads * 4, dtype=torch.int32, device=dev)
            cnt = torch.zeros(4, dtype=torch.int64, device=dev)  # fragments, classified, bases, lookups
            eng.classify_device(bases.data_ptr(), offs.data_ptr(), n_reads, False, 0.0, out.data_ptr(), cnt.data_ptr())
            torch.cuda.synchronize()
            rec = out.cpu().numpy().view(np.uint32).reshape(n_reads, 4)
            assert cnt[0].item() == n_reads and cnt[3].item() > 30 * n_reads
            res[name] = (rec.copy(), cnt[3].item())
            assert (rec[: n_reads // 2, 0] == 30).all(), name  # every read of inserted sequence is found
            assert (rec[: n_reads // 2, 2] == R - 35 + 1).all(), name  # all of its k-mers hit the taxon
            # random reads: only chance matches of the compacted keys, far below 1 %
            assert (rec[n_reads // 2:, 0] != 0).mean() < 0.01, name
    assert res["small"][1] == res["wide"][1]  # same minimizer runs -> same number of lookups
    assert np.array_equal(res["small"][0][: n_reads // 2], res["wide"][0][: n_reads // 2])


@pytest.mark.parametrize("value_bits", [8, 16, 24, 28])
def test_wide_values_and_short_keys_collide_identically(value_bits):
    """The cell layout is key_bits + value_bits = 32.  With wide values only a few key bits are left
    (4 at value_bits 28), so unrelated minimizers match by chance all the time: the GPU must report
    exactly the oracle's (false) hits, LCAs over a 200-level deep taxonomy included."""
    from nohuman_amd import Engine
    from oracle import minidb
    rng = np.random.default_rng(100 + value_bits)
    edges = {1: 0}
    for d in range(2, 202):      # a chain 200 levels deep ...
        edges[d] = d - 1
    leaves = []
    for i in range(40):          # ... with leaves hanging off it at many depths
        ext = 1000 + i
        edges[ext] = 5 * i + 2
        leaves.append(ext)
    tax = minidb.Taxonomy(edges)
    segs = {ext: synth.random_seq(rng, 400) for ext in leaves}
    hashb, size = minidb.build_hash(tax, sorted(segs.items()), 30011, value_bits=value_bits)
    ob, tb = minidb.opts_bytes(), tax.to_bytes()
    reads = synth.sample_reads(rng, segs, 1500, length=150, frac_random=0.5)
    reads += [b"".join(segs[int(e)][50:250] for e in rng.choice(leaves, size=6, replace=False)) for _ in range(60)]
    bases, offs = orc.pack_reads(reads, False)
    odb = orc.OracleDB(ob, tb, hashb)
    for conf in (0.0, 0.2):
        exp, lookups, etaxa, _ = odb.classify(bases, offs, False, conf, want_taxa=True)
        with Engine.from_images(ob, tb, hashb) as eng:
            assert eng.info.value_bits == value_bits
            got, taxa, _ = eng.classify(bases, offs, False, conf, want_taxa=True)
        _assert_same(got, exp, "value_bits=%d conf=%s" % (value_bits, conf))
        assert np.array_equal(taxa, etaxa)
    if value_bits >= 24:  # chance matches really happen: random reads get classified
        assert (exp["call"][750:1500] != 0).mean() > 0.02


@pytest.mark.parametrize("seed", range(40))
def test_random_database_geometries(seed):
    """A seeded sweep over what opts.k2d / hash.k2d can say -- k, l, spaced seed, toggle, legacy
    reverse complement, min-hash subsampling, load factor, probing rule, paired or not, confidence,
    mate reset, hit-group threshold -- with reads around every length boundary: GPU == oracle."""
    from nohuman_amd import Engine
    rng = np.random.default_rng(7000 + seed)
    l = int(rng.integers(8, 32))
    k = int(l + rng.integers(0, 12))
    kw = dict(k=k, l=l)
    if rng.random() < 0.5:  # spaced seed: clear the low bit of some of the last base pairs
        s = int(rng.integers(1, max(2, l // 4)))
        mask = (1 << (2 * l)) - 1
        for i in range(s):
            mask &= ~(0b10 << (4 * i))
        kw["spaced_mask"] = mask
    else:
        kw["spaced_mask"] = 0
    kw["toggle"] = int(rng.integers(0, 1 << 62))
    kw["revcom_version"] = int(rng.integers(0, 2))
    if rng.random() < 0.3:
        kw["min_hash"] = int(rng.integers(1 << 60, 1 << 63))
    linear = bool(rng.random() < 0.7)
    kw["linear_probing"] = linear
    capacity = int(rng.choice([1999, 2503, 4001, 9973]))  # load factor ~0.9 ... 0.2
    ob, tb, hb, genomes, _ = synth.toy_db(seed=seed, seg=int(rng.integers(150, 400)), capacity=capacity, **kw)
    odb = orc.OracleDB(ob, tb, hb)
    paired = bool(rng.random() < 0.5)
    reads = synth.sample_reads(rng, genomes, 400, length=int(rng.integers(k, 3 * k + 40)), paired=paired,
                               len_jitter=int(rng.integers(0, k)), n_rate=float(rng.choice([0.0, 0.002, 0.02])))
    g = genomes[sorted(genomes)[0]]
    edge = [g[:n] for n in (0, 1, l - 1, l, k - 1, k, k + 1, k + 123, k + 124, k + 125, 2 * 124 + k)]
    reads += [(a, b) for a, b in zip(edge, edge[::-1])] if paired else edge
    bases, offs = orc.pack_reads(reads, paired)
    conf = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    opts = dict(linear_probing=linear, reset_per_mate=bool(rng.random() < 0.8),
                minimum_hit_groups=int(rng.integers(0, 4)))
    with Engine.from_images(ob, tb, hb) as eng:
        odb.set(**opts)
        eng.set_options(**opts)
        exp, lookups, etaxa, _ = odb.classify(bases, offs, paired, conf, want_taxa=True)
        eng.reset_stats()
        got, taxa, _ = eng.classify(bases, offs, paired, conf, want_taxa=True)
        st = eng.stats()
    _assert_same(got, exp, "seed %d k=%d l=%d %s" % (seed, k, l, kw))
    assert np.array_equal(taxa, etaxa)
    assert st.table_lookups == int(lookups.sum())


@pytest.mark.parametrize("rule", [1, 0])
@pytest.mark.parametrize("k,l", [(19, 8), (35, 15), (31, 10), (72, 8), (40, 19)])
def test_window_wider_than_lmer_with_ambiguous_bases(k, l, rule):
    """k - l > l: the (k-l+1)-window of a k-mer reaches l-mers that lie wholly BEFORE an ambiguous base
    whose own l-mers are long past.  kraken2's scanner cleared its queue at that base, so they must not
    take part in the minimum (the kernel used a plain min over the window)."""
    from nohuman_amd import Engine
    rng = np.random.default_rng(k * 100 + l)
    ob, tb, hb, genomes, _ = synth.toy_db(seed=k + l, k=k, l=l, spaced_mask=0, capacity=9973)
    odb = orc.OracleDB(ob, tb, hb)
    odb.set(ambiguity_rule=rule)
    for paired in (False, True):
        reads = synth.sample_reads(rng, genomes, 500, length=160, paired=paired, len_jitter=60, n_rate=0.02,
                                   frac_random=0.2)
        bases, offs = orc.pack_reads(reads, paired)
        exp, lookups, etaxa, _ = odb.classify(bases, offs, paired, 0.05, want_taxa=True)
        with Engine.from_images(ob, tb, hb) as eng:
            eng.set_options(ambiguity_rule=rule)
            got, taxa, _ = eng.classify(bases, offs, paired, 0.05, want_taxa=True)
            st = eng.stats()
        _assert_same(got, exp, "k=%d l=%d paired=%s" % (k, l, paired))
        assert np.array_equal(taxa, etaxa)
        assert st.table_lookups == int(lookups.sum())


@pytest.mark.parametrize("paired", [False, True])
def test_sequences_in_place_inside_their_record_text(toy, toy_oracle, paired):
    """nh_classify_records_device: the sequences stay where the reader put them -- inside FASTQ text with
    headers of every length (so the sequences start at every byte alignment), '+' lines, qualities that
    look like bases ('A', 'N', '@'), CRLF -- and the kernel gets (start, length) per sequence.  Mixed
    lengths: chunks of short reads go through the short-read kernel, chunks holding a longer read are
    left to the generic one.  Records, per-k-mer taxa and counters equal the oracle's on the packed reads."""
    import torch
    from nohuman_amd import Engine
    ob, tb, hb, genomes, _ = toy
    rng = np.random.default_rng(31 + paired)
    reads = synth.sample_reads(rng, genomes, 3000, paired=paired, len_jitter=60, n_rate=0.004)
    longr = synth.sample_reads(rng, genomes, 40, length=700, paired=paired, len_jitter=300, n_rate=0.002)
    order = rng.permutation(len(reads) + len(longr))
    allr = [(reads + longr)[i] for i in order]
    allr[7] = (b"", b"ACGT") if paired else b""  # no k-mer at all
    seqs = [s for fr in allr for s in (fr if paired else (fr,))]
    text = bytearray()
    starts, lens = [], []
    texts = [bytearray(), bytearray()]
    pos = [[], []]
    for i, fr in enumerate(allr):
        for m, s in enumerate(fr if paired else (fr,)):
            t = texts[m]
            t += b"@r%d%s/%d\n" % (i, b"x" * int(rng.integers(0, 9)), m + 1)
            pos[m].append((len(t), len(s)))
            q = bytes(rng.choice(np.frombuffer(b"ANI@#ACGT", dtype=np.uint8), size=len(s)))
            t += s + (b"\r\n+\r\n" if i % 50 == 0 else b"\n+\n") + q + b"\n"
    base2 = (len(texts[0]) + 8 + 255) & ~255
    blob = bytes(texts[0]) + b"\0" * (base2 - len(texts[0])) + (bytes(texts[1]) if paired else b"")
    for i in range(len(allr)):
        for m in range(2 if paired else 1):
            st, ln = pos[m][i]
            starts.append(st + (base2 if m else 0))
            lens.append(ln)
    bases, offs = orc.pack_reads(allr, paired)
    exp, lookups, etaxa, etoff = toy_oracle.classify(bases, offs, paired, 0.05, want_taxa=True)
    dev = torch.device("cuda:0")
    d_text = torch.frombuffer(bytearray(blob + b"\0" * 64), dtype=torch.uint8).to(dev)
    d_st = torch.tensor(starts, dtype=torch.int64, device=dev)
    d_ln = torch.tensor(lens, dtype=torch.int32, device=dev)
    n = len(allr)
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    taxa = torch.zeros(len(etaxa) + 1, dtype=torch.int32, device=dev)
    toff = torch.tensor(np.asarray(etoff, dtype=np.int64), device=dev)
    with Engine.from_images(ob, tb, hb) as eng:
        eng.classify_records_device(d_text.data_ptr(), len(blob), d_st.data_ptr(), d_ln.data_ptr(), n, paired, 0.05,
                                    out.data_ptr(), cnt.data_ptr(), 0, taxa.data_ptr(), toff.data_ptr())
        torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    for i, f in enumerate(("call", "total_kmers", "clade_hits", "hit_groups")):
        bad = np.nonzero(got[:, i] != exp[f])[0]
        assert bad.size == 0, "%s differs at %s" % (f, bad[:5])
    assert np.array_equal(taxa.cpu().numpy().view(np.uint32)[: len(etaxa)], etaxa)
    c = cnt.tolist()
    assert c == [n, int((exp["call"] != 0).sum()), sum(len(s) for s in seqs), int(lookups.sum())]


@pytest.mark.parametrize("capacity", [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 47, 64, 100, 257])
def test_tiny_tables_wrap_around_and_full_cycles(toy, capacity):
    """Tables smaller than one probe round, one line, two lines: probe runs wrap around the end of the table
    (several times inside one 16-cell quad round), and on a FULL table a miss must stop after exactly one
    cycle.  Cells are random (taxa of the toy tree), so every lookup is a miss or a chance
    match (6 key bits): GPU == oracle on records, per-k-mer taxa and lookup counts, single copy and staggered copies."""
    import struct
    from nohuman_amd import Engine
    ob, tb, _, genomes, tax = toy
    vb = 26  # 6 key bits: unrelated minimizers match by chance all the time
    rng = np.random.default_rng(capacity)
    for fill in (0.5, 0.9, 1.0):
        cells = np.zeros(capacity, dtype=np.uint32)
        occ = rng.random(capacity) < fill if fill < 1.0 else np.ones(capacity, bool)
        keys = rng.integers(0, 1 << 6, size=capacity, dtype=np.uint32)
        vals = rng.integers(1, 10, size=capacity, dtype=np.uint32)
        cells[occ] = ((keys[occ] << vb) | vals[occ]).astype(np.uint32)
        hb = struct.pack("<4Q", capacity, int(occ.sum()), 32 - vb, vb) + cells.tobytes()
        reads = [synth.random_seq(rng, int(n)) for n in rng.integers(35, 200, size=300)]
        bases, offs = orc.pack_reads(reads, False)
        odb = orc.OracleDB(ob, tb, hb)
        exp, lookups, etaxa, _ = odb.classify(bases, offs, False, 0.0, want_taxa=True)
        for copies in ("1", "4"):
            os.environ["NOHUMAN_TABLE_COPIES"] = copies
            try:
                with Engine.from_images(ob, tb, hb) as eng:
                    got, taxa, _ = eng.classify(bases, offs, False, 0.0, want_taxa=True)
                    st = eng.stats()
            finally:
                os.environ.pop("NOHUMAN_TABLE_COPIES", None)
            _assert_same(got, exp, "capacity %d fill %.1f copies %s" % (capacity, fill, copies))
            assert np.array_equal(taxa, etaxa)
            assert st.table_lookups == int(lookups.sum())


@pytest.mark.parametrize("seed", range(16))
def test_default_geometry_length_mixtures_around_the_tile_boundary(toy, toy_oracle, seed):
    """Default k=35/l=31 database, reads whose lengths straddle every boundary the two kernels care about:
    no k-mer (< 35), one tile exactly (158), one base more (159: the chunk is left to the generic kernel),
    multi-tile reads, empty mates -- shuffled, so that chunks of 24 pairs / 32 reads mix short-only and
    deferred chunks; paired or not, with N and lower case, several confidences and chunk sizes."""
    from nohuman_amd import Engine
    ob, tb, hb, genomes, _ = toy
    rng = np.random.default_rng(9000 + seed)
    paired = bool(seed & 1)
    pools = [np.arange(0, 40), np.arange(150, 171), np.array([157, 158, 159, 160]), np.arange(280, 330),
             np.arange(600, 1400, 37)]
    weights = [0.1, 0.45, 0.2, 0.15, 0.1] if seed % 4 else [0.1, 0.7, 0.2, 0.0, 0.0]  # some seeds: short reads only
    n = 1500
    lens = [int(rng.choice(pools[int(rng.choice(5, p=weights))])) for _ in range(n * (2 if paired else 1))]
    g = sorted(genomes)
    seqs = []
    for ln in lens:
        src = genomes[g[int(rng.integers(0, len(g)))]]
        if rng.random() < 0.3 or len(src) <= ln:
            s = synth.random_seq(rng, ln)
        else:
            st = int(rng.integers(0, len(src) - ln))
            s = src[st:st + ln]
            if rng.random() < 0.5:
                s = synth.revcomp(s)
        seqs.append(synth.mutate(rng, s, 0.01, float(rng.choice([0.0, 0.003, 0.03])), 0.05))
    reads = list(zip(seqs[0::2], seqs[1::2])) if paired else seqs
    bases, offs = orc.pack_reads(reads, paired)
    conf = float(rng.choice([0.0, 0.1, 0.6]))
    exp, lookups, etaxa, _ = toy_oracle.classify(bases, offs, paired, conf, want_taxa=True)
    chunk = str(int(rng.choice([1, 3, 7, 24, 31]))) if paired else str(int(rng.choice([1, 5, 32, 63])))
    os.environ["NOHUMAN_FRAG_CHUNK"] = chunk
    try:
        with Engine.from_images(ob, tb, hb) as eng:
            got, taxa, _ = eng.classify(bases, offs, paired, conf, want_taxa=True)
            st = eng.stats()
    finally:
        os.environ.pop("NOHUMAN_FRAG_CHUNK", None)
    _assert_same(got, exp, "seed %d paired %s conf %s chunk %s" % (seed, paired, conf, chunk))
    assert np.array_equal(taxa, etaxa)
    assert st.table_lookups == int(lookups.sum()) and st.total_bases == sum(lens)


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("hidden_short", [False, True])
def test_reads_longer_than_a_tile_throughout_skip_the_short_read_kernel(toy, toy_oracle, paired, hidden_short):
    """2 x 250 bp: every one of the 64 sequences both kernels sample is longer than a tile (158 bases), so
    k_classify_short returns at once and the generic kernel classifies every chunk (sample_all_long, nh_kernels.hip).
    hidden_short: short reads (and reads without a k-mer) at places the sample does not look at -- the shortcut is a
    scheduling decision, the results must not depend on what the sample saw."""
    from nohuman_amd import Engine
    ob, tb, hb, genomes, _ = toy
    rng = np.random.default_rng(77 + 2 * paired + hidden_short)
    n = 1300
    ns = n * (2 if paired else 1)
    lens = [int(rng.integers(200, 320)) for _ in range(ns)]
    if hidden_short:
        sampled = {ns * i // 64 for i in range(64)}
        for j in rng.choice(ns, 200, replace=False):
            if int(j) not in sampled:
                lens[int(j)] = int(rng.choice([0, 20, 35, 100, 150, 158]))
    g = sorted(genomes)
    seqs = []
    for ln in lens:
        src = genomes[g[int(rng.integers(0, len(g)))]]
        st = int(rng.integers(0, len(src) - ln))
        seqs.append(synth.mutate(rng, src[st:st + ln], 0.01, 0.003, 0.0) if ln else b"")
    reads = list(zip(seqs[0::2], seqs[1::2])) if paired else seqs
    bases, offs = orc.pack_reads(reads, paired)
    exp, lookups, etaxa, _ = toy_oracle.classify(bases, offs, paired, 0.05, want_taxa=True)
    with Engine.from_images(ob, tb, hb) as eng:
        got, taxa, _ = eng.classify(bases, offs, paired, 0.05, want_taxa=True)
        st = eng.stats()
    _assert_same(got, exp, "paired %s hidden_short %s" % (paired, hidden_short))
    assert np.array_equal(taxa, etaxa)
    assert st.table_lookups == int(lookups.sum()) and st.total_bases == sum(lens) and st.total_sequences == n
