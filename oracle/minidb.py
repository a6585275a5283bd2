"""
minidb.py -- TEST-ONLY writer of tiny, valid kraken2 databases (opts.k2d / taxo.k2d / hash.k2d).

The reference tree holds no .k2d sample (its tests use empty files of those names,
/root/reference/src/download.rs:507-549), and HPRC.r2 cannot be downloaded here, so hit / LCA /
confidence paths are exercised on databases built by this file from synthetic genomes and a toy
taxonomy.  Build semantics restate kraken2's build_db.cc ProcessSequence + CompactHashTable::
CompareAndSet: every non-ambiguous k-mer's minimizer is inserted with
value = LCA(existing value, taxon).  File layouts follow SURVEY.md A.1.

TEST INFRASTRUCTURE ONLY -- nothing under nohuman_amd/ may import this.
"""
from __future__ import annotations

import struct

from . import k2_literal as lit

DEFAULT_K, DEFAULT_L = 35, 31
DEFAULT_TOGGLE = 0xE37E28C4271B5A2D


def default_spaced_mask(l: int = DEFAULT_L, spaces: int = 7) -> int:
    """2-bit-per-base expansion of "1"*(l-2s) + "01"*s."""
    pattern = "1" * (l - 2 * spaces) + "01" * spaces
    mask = 0
    for ch in pattern:
        mask = (mask << 2) | (3 if ch == "1" else 0)
    return mask


class Taxonomy:
    """Toy taxonomy: external ids + parent links -> BFS internal ids (node 0 = null sentinel)."""

    def __init__(self, edges: dict, names: dict | None = None):
        # edges: external child id -> external parent id; the root maps to 0
        self.edges = dict(edges)
        roots = [c for c, p in edges.items() if p == 0]
        assert len(roots) == 1
        order = [roots[0]]
        i = 0
        while i < len(order):
            kids = sorted(c for c, p in edges.items() if p == order[i])
            order.extend(kids)
            i += 1
        self.external = [0] + order
        self.internal = {e: i for i, e in enumerate(self.external)}
        self.parent = [0] * len(self.external)
        for e, p in edges.items():
            self.parent[self.internal[e]] = self.internal[p] if p else 0
        self.names = names or {}

    @property
    def node_count(self) -> int:
        return len(self.external)

    def lca(self, a: int, b: int) -> int:
        if not a or not b:
            return a or b
        while a != b:
            if a > b:
                a = self.parent[a]
            else:
                b = self.parent[b]
        return a

    def to_bytes(self) -> bytes:
        n = self.node_count
        name_blob = bytearray()
        rank_blob = bytearray(b"no rank\0")
        first_child = [0] * n
        child_count = [0] * n
        for i in range(1, n):
            p = self.parent[i]
            if p:
                if child_count[p] == 0:
                    first_child[p] = i
                child_count[p] += 1
        nodes = bytearray()
        for i in range(n):
            name_off = len(name_blob)
            name_blob += (self.names.get(self.external[i], "taxon%d" % self.external[i])).encode() + b"\0"
            nodes += struct.pack("<7Q", self.parent[i], first_child[i], child_count[i], name_off, 0,
                                 self.external[i], 0)
        return (b"K2TAXDAT" + struct.pack("<3Q", n, len(name_blob), len(rank_blob)) + bytes(nodes)
                + bytes(name_blob) + bytes(rank_blob))


def opts_bytes(k=DEFAULT_K, l=DEFAULT_L, spaced_mask=None, toggle=DEFAULT_TOGGLE, dna_db=1,
               min_hash=0, revcom_version=1, db_version=0, db_type=0) -> bytes:
    if spaced_mask is None:
        spaced_mask = default_spaced_mask(l) if l == DEFAULT_L else 0
    return (struct.pack("<4Q", k, l, spaced_mask, toggle) + struct.pack("<B7x", dna_db)
            + struct.pack("<Q", min_hash) + struct.pack("<3i4x", revcom_version, db_version, db_type))


def value_bits_for(node_count: int) -> int:
    b = 1
    while (1 << b) < node_count:
        b += 1
    return b


def build_hash(taxonomy: Taxonomy, genomes, capacity: int, *, k=DEFAULT_K, l=DEFAULT_L,
               spaced_mask=None, toggle=DEFAULT_TOGGLE, revcom_version=1, value_bits=None,
               linear_probing=True, min_hash=0, ambiguity_rule=0):
    """genomes: iterable of (external_taxid, bytes).  Returns (hash_bytes, size).
    ambiguity_rule: which k-mers next to an ambiguous base the builder skips (k2_literal.DB.ambiguity_rule); 0 keeps
    the committed toy database byte-stable -- a table is a table, whichever rule its builder followed."""
    if spaced_mask is None:
        spaced_mask = default_spaced_mask(l) if l == DEFAULT_L else 0
    vb = value_bits if value_bits is not None else value_bits_for(taxonomy.node_count)
    kb = 32 - vb
    vmask = (1 << vb) - 1
    cells = [0] * capacity
    size = 0
    shim = lit.DB(k, l, spaced_mask, toggle, 1, min_hash, revcom_version, capacity, 0, kb, vb,
                  cells, taxonomy.parent, taxonomy.external, linear_probing=linear_probing,
                  ambiguity_rule=ambiguity_rule)
    for ext, seq in genomes:
        taxon = taxonomy.internal[ext]
        for ambiguous, minimizer in lit.kmer_minimizers(shim, seq):
            if ambiguous:
                continue
            hc = lit.fmix64(minimizer)
            if min_hash and hc < min_hash:
                continue
            compacted = hc >> (32 + vb)
            idx = hc % capacity
            first = idx
            step = 1 if linear_probing else ((hc >> 8) | 1)
            while True:
                cell = cells[idx]
                if cell & vmask == 0:
                    cells[idx] = (compacted << vb) | taxon
                    size += 1
                    break
                if cell >> vb == compacted:
                    cells[idx] = (compacted << vb) | taxonomy.lca(cell & vmask, taxon)
                    break
                idx = (idx + step) % capacity
                assert idx != first, "mini-DB hash table full"
    blob = struct.pack("<4Q", capacity, size, kb, vb) + struct.pack("<%dI" % capacity, *cells)
    return blob, size


def write_db(dirpath, opts: bytes, taxo: bytes, hashb: bytes):
    import os
    os.makedirs(dirpath, exist_ok=True)
    for name, blob in (("opts.k2d", opts), ("taxo.k2d", taxo), ("hash.k2d", hashb)):
        with open(os.path.join(dirpath, name), "wb") as f:
            f.write(blob)
