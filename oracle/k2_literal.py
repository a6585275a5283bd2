"""
k2_literal.py -- second, independent CPU restatement of the kraken2 classify path.
TEST INFRASTRUCTURE ONLY (see oracle/k2_oracle.h for the parity statement: PARITY UNPINNED vs
kraken2; nothing under nohuman_amd/ may import this).

Where oracle/k2_oracle.c follows kraken2's *state machine* (mmscanner.cc NextMinimizer with its
monotone deque), this file states the same result in *closed form* per k-mer (SURVEY.md
Appendix A.3) with pure-Python integers and dicts, so the two can be differential-tested.  It is
also the generator of the committed fixtures in tests/golden/ (tests/golden/make_golden.py).

Reference call sites of the path: /root/reference/src/main.rs:215-270, src/lib.rs:22-48;
kraken2 pin: /root/reference/Dockerfile:15,35-38.
"""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass, field

AMBIG = 0xFFFFFFFF  # "A" spans in the hit list
BORDER = 0xFFFFFFFE  # "|:|" mate border
M64 = (1 << 64) - 1

_CODE = {ord("A"): 0, ord("a"): 0, ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2,
         ord("T"): 3, ord("t"): 3}


def fmix64(k: int) -> int:
    """kv_store.h MurmurHash3 (A.4)."""
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & M64
    k ^= k >> 33
    return k


def reverse_complement(x: int, n: int, revcom_version: int = 1) -> int:
    """mmscanner.cc reverse_complement (A.2), stated base by base instead of by bit swaps."""
    if revcom_version == 0:
        # legacy bug: the 64-bit word is reversed/complemented but not shifted down
        word = 0
        for i in range(32):
            base = (x >> (2 * i)) & 3
            word |= (3 - base) << (2 * (31 - i))
        return word & ((1 << (2 * n)) - 1)
    out = 0
    for i in range(n):
        base = (x >> (2 * i)) & 3
        out = (out << 2) | (3 - base)
    return out


@dataclass
class DB:
    k: int
    l: int
    spaced_seed_mask: int
    toggle_mask: int
    dna_db: int
    min_hash: int
    revcom_version: int
    capacity: int
    size: int
    key_bits: int
    value_bits: int
    cells: list
    parent: list
    external: list
    linear_probing: bool = True
    reset_per_mate: bool = True
    minimum_hit_groups: int = 2
    # 0: a k-mer is ambiguous iff an ambiguous byte lies in its last l bases; 1 (default): mmscanner.h
    # is_ambiguous() = queue_pos < k-l || last_ambig, in closed form: in its last max(l, k-1) bases
    ambiguity_rule: int = int(os.environ.get("K2O_AMBIGUITY_RULE", "1") != "0")
    names: list = field(default_factory=list)

    @classmethod
    def from_images(cls, opts: bytes, taxo: bytes, hashb: bytes) -> "DB":
        ob = (opts + b"\0" * 64)[:64] if len(opts) < 64 else opts[:64]
        k, l, mask, toggle = struct.unpack_from("<4Q", ob, 0)
        dna = ob[32]
        (min_hash,) = struct.unpack_from("<Q", ob, 40)
        (rv,) = struct.unpack_from("<i", ob, 48)
        cap, size, kb, vb = struct.unpack_from("<4Q", hashb, 0)
        assert len(hashb) == 32 + 4 * cap
        cells = list(struct.unpack_from("<%dI" % cap, hashb, 32))
        assert taxo[:8] == b"K2TAXDAT"
        nc, nl, rl = struct.unpack_from("<3Q", taxo, 8)
        assert len(taxo) == 32 + 56 * nc + nl + rl
        parent, ext = [], []
        for i in range(nc):
            f = struct.unpack_from("<7Q", taxo, 32 + 56 * i)
            parent.append(f[0])
            ext.append(f[5])
        return cls(k, l, mask, toggle, dna, min_hash, rv, cap, size, kb, vb, cells, parent, ext)

    # -- compact_hash.cc Get (A.4)
    def get(self, minimizer: int) -> int:
        hc = fmix64(minimizer)
        vmask = (1 << self.value_bits) - 1
        compacted = hc >> (32 + self.value_bits)
        idx = hc % self.capacity
        first = idx
        step = 1 if self.linear_probing else ((hc >> 8) | 1)
        while True:
            cell = self.cells[idx]
            if cell & vmask == 0:
                return 0
            if cell >> self.value_bits == compacted:
                return cell & vmask
            idx = (idx + step) % self.capacity
            if idx == first:
                return 0

    # -- taxonomy.cc (A.5)
    def is_a_ancestor_of_b(self, a: int, b: int) -> bool:
        if not a or not b:
            return False
        while b > a:
            b = self.parent[b]
        return a == b

    def lca(self, a: int, b: int) -> int:
        if not a or not b:
            return a or b
        while a != b:
            if a > b:
                a = self.parent[a]
            else:
                b = self.parent[b]
        return a


def kmer_minimizers(db: DB, seq: bytes):
    """Closed form of A.3: list of (ambiguous, minimizer) per k-mer end position."""
    k, l = db.k, db.l
    n = len(seq)
    lmask = (1 << (2 * l)) - 1
    toggle = db.toggle_mask & lmask
    codes = [_CODE.get(c, -1) for c in seq]
    # candidate (toggled, masked, canonical) value of the l-mer ENDING at j, None if it holds
    # an ambiguous byte
    cand = [None] * n
    for j in range(l - 1, n):
        window = codes[j - l + 1: j + 1]
        if min(window) < 0:
            continue
        x = 0
        for c in window:
            x = (x << 2) | c
        canon = min(x, reverse_complement(x, l, db.revcom_version))
        if db.spaced_seed_mask:
            canon &= db.spaced_seed_mask
        cand[j] = canon ^ toggle
    out = []
    span = l if db.ambiguity_rule == 0 else max(l, k - 1)
    for e in range(k - 1, n):
        # last ambiguous byte at or before e
        p = -1
        for i in range(e, -1, -1):
            if codes[i] < 0:
                p = i
                break
        if p > e - span:  # ambiguous byte within the last `span` bases
            out.append((True, None))
            continue
        lo = max(e - (k - l), p + l)
        best = min(cand[j] for j in range(lo, e + 1))
        out.append((False, best ^ toggle))
    return out


def resolve_tree(db: DB, hit_counts: dict, total_kmers: int, confidence: float):
    """classify.cc ResolveTree (A.5), order-independent statement."""
    required = int(math.ceil(confidence * total_kmers)) & 0xFFFFFFFF
    scores = {}
    for t in hit_counts:
        scores[t] = sum(c for u, c in hit_counts.items() if db.is_a_ancestor_of_b(u, t))
    best = 0
    if scores:
        top = max(scores.values())
        for t, s in scores.items():
            if s == top:
                best = db.lca(best, t)
    score = hit_counts.get(best, 0)
    while best and score < required:
        score = sum(c for u, c in hit_counts.items() if db.is_a_ancestor_of_b(best, u))
        if score >= required:
            break
        best = db.parent[best]
    clade = sum(c for u, c in hit_counts.items() if db.is_a_ancestor_of_b(best, u)) if best else 0
    return best, clade


def classify_fragment(db: DB, mates, confidence: float):
    """classify.cc ClassifySequence, nucleotide branch (A.5).
    Returns (call, total_kmers, clade_hits, hit_groups, taxa_list, lookups)."""
    paired = len(mates) == 2
    hit_counts: dict = {}
    taxa = []
    hit_groups = 0
    lookups = 0
    last_min, last_taxon = None, None
    for mi, seq in enumerate(mates):
        if db.reset_per_mate:
            last_min, last_taxon = None, None
        for ambiguous, minimizer in kmer_minimizers(db, seq):
            if ambiguous:
                taxa.append(AMBIG)
                continue
            if minimizer != last_min:
                taxon = 0
                if not (db.min_hash and fmix64(minimizer) < db.min_hash):
                    taxon = db.get(minimizer)
                    lookups += 1
                last_min, last_taxon = minimizer, taxon
                if taxon:
                    hit_groups += 1
            else:
                taxon = last_taxon
            if taxon:
                hit_counts[taxon] = hit_counts.get(taxon, 0) + 1
            taxa.append(taxon)
        if paired and mi == 0:
            taxa.append(BORDER)
    total_kmers = len(taxa) - (1 if paired else 0)
    call, clade = resolve_tree(db, hit_counts, total_kmers, confidence)
    if call and hit_groups < db.minimum_hit_groups:
        call, clade = 0, 0
    return call, total_kmers, clade, hit_groups, taxa, lookups


def hitlist_string(db: DB, taxa) -> str:
    """classify.cc AddHitlistString (A.6): run-length `extid:count`, `A:n`, `|:|`, `0:0` if empty."""
    if not taxa:
        return "0:0"
    parts = []
    i = 0
    while i < len(taxa):
        t = taxa[i]
        j = i
        while j < len(taxa) and taxa[j] == t:
            j += 1
        if t == BORDER:
            parts.extend(["|:|"] * (j - i))
        elif t == AMBIG:
            parts.append("A:%d" % (j - i))
        else:
            parts.append("%d:%d" % (db.external[t], j - i))
        i = j
    return " ".join(parts)
