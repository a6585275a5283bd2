"""Deterministic synthetic inputs for the tests (SURVEY.md section 8d generator conventions)."""
from __future__ import annotations

import numpy as np

from oracle import minidb

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_seq(rng: np.random.Generator, n: int) -> bytes:
    return ACGT[rng.integers(0, 4, size=n)].tobytes()


def mutate(rng: np.random.Generator, seq: bytes, sub_rate=0.0, n_rate=0.0, lower_rate=0.0) -> bytes:
    a = np.frombuffer(seq, dtype=np.uint8).copy()
    if sub_rate:
        m = rng.random(a.size) < sub_rate
        a[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    if lower_rate:
        m = rng.random(a.size) < lower_rate
        a[m] |= 0x20
    if n_rate:
        m = rng.random(a.size) < n_rate
        a[m] = ord("N")
    return a.tobytes()


def revcomp(seq: bytes) -> bytes:
    return seq.translate(bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN"))[::-1]


# toy taxonomy: external ids; root 1
TOY_EDGES = {1: 0, 10: 1, 20: 1, 11: 10, 12: 10, 111: 11, 112: 11, 21: 20, 9606: 20}


def toy_db(seed=7, seg=400, capacity=4001, **opts_kw):
    """Genomes share segments along the tree so LCA values at every depth appear in the table.
    Returns (opts, taxo, hash, genomes dict ext->bytes, Taxonomy)."""
    rng = np.random.default_rng(seed)
    tax = minidb.Taxonomy(TOY_EDGES)
    segs = {e: random_seq(rng, seg) for e in TOY_EDGES}

    def lineage(e):
        out = []
        while e:
            out.append(e)
            e = TOY_EDGES[e]
        return out[::-1]

    leaves = [e for e in TOY_EDGES if e not in TOY_EDGES.values()]
    genomes = {}
    for leaf in leaves:
        # shared ancestor segments + an own segment; 'N' spacer keeps k-mers from spanning segments
        genomes[leaf] = b"N".join(segs[a] for a in lineage(leaf))
    k = opts_kw.get("k", minidb.DEFAULT_K)
    l = opts_kw.get("l", minidb.DEFAULT_L)
    build_kw = {kk: vv for kk, vv in opts_kw.items()
                if kk in ("k", "l", "spaced_mask", "toggle", "revcom_version", "min_hash")}
    linear = opts_kw.get("linear_probing", True)
    hashb, size = minidb.build_hash(tax, sorted(genomes.items()), capacity, linear_probing=linear,
                                    **build_kw)
    ob = minidb.opts_bytes(k=k, l=l, spaced_mask=opts_kw.get("spaced_mask"),
                           toggle=opts_kw.get("toggle", minidb.DEFAULT_TOGGLE),
                           min_hash=opts_kw.get("min_hash", 0),
                           revcom_version=opts_kw.get("revcom_version", 1))
    return ob, tax.to_bytes(), hashb, genomes, tax


def sample_reads(rng, genomes, n, length=150, paired=False, sub_rate=0.01, n_rate=0.002,
                 lower_rate=0.05, frac_random=0.3, len_jitter=0):
    """Reads drawn from the toy genomes (both strands) mixed with uniform-random reads."""
    keys = sorted(genomes)
    reads = []
    for _ in range(n):
        def one():
            ln = length + (int(rng.integers(-len_jitter, len_jitter + 1)) if len_jitter else 0)
            ln = max(0, ln)
            if rng.random() < frac_random:
                s = random_seq(rng, ln)
            else:
                g = genomes[keys[int(rng.integers(0, len(keys)))]]
                if len(g) <= ln:
                    s = g
                else:
                    st = int(rng.integers(0, len(g) - ln))
                    s = g[st:st + ln]
                if rng.random() < 0.5:
                    s = revcomp(s)
            return mutate(rng, s, sub_rate, n_rate, lower_rate)
        reads.append((one(), one()) if paired else one())
    return reads
