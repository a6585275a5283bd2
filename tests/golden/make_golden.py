#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ with the closed-form Python restatement
(oracle/k2_literal.py) -- NOT with the C oracle or the GPU path, which are checked against them.

    python tests/golden/make_golden.py

Outputs (all small):
  toy_db/{opts,taxo,hash}.k2d    mini database written by oracle/minidb.py (toy taxonomy)
  reads_se.fq, reads_pe_1.fq, reads_pe_2.fq   synthetic reads incl. N, lower case, short reads
  expected_se.json, expected_pe.json          per-fragment records, kraken hit lists, lookups D under the default
                                              ambiguity rule (1: mmscanner.h is_ambiguous(), an N costs k-1 k-mers)
  expected_se_rule0.json, expected_pe_rule0.json   the same under rule 0 (ambiguous byte in the last l bases: A:31)
  kat.json                                    known answers for fmix64 / reverse complement / masks
There is no reference-held vector for this path (parity unpinned); these
fixtures pin the two independent restatements and the HIP kernels to each other.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import k2_literal as lit  # noqa: E402
from oracle import minidb  # noqa: E402
from tests import synth  # noqa: E402

CONFS = [0.0, 0.1, 0.5]


def write_fastq(path, reads, tag, mate=None):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            name = b"@%s.%d" % (tag.encode(), i)
            if mate:
                name += b"/%d" % mate
            # every 7th record carries a description and a '+id' line (normalised by the writer)
            desc = b" desc=%d\tx" % i if i % 7 == 0 else b""
            plus = b"+" + name[1:] if i % 7 == 0 else b"+"
            f.write(name + desc + b"\n" + r + b"\n" + plus + b"\n" + b"I" * len(r) + b"\n")


def expected(db, frags, paired):
    out = []
    for fr in frags:
        mates = fr if paired else (fr,)
        rec = {"len": [len(m) for m in mates], "by_conf": {}}
        for conf in CONFS:
            call, tk, ch, hg, taxa, lookups = lit.classify_fragment(db, mates, conf)
            rec["by_conf"][str(conf)] = [call, tk, ch, hg]
        rec["hitlist"] = lit.hitlist_string(db, taxa)
        rec["lookups"] = lookups
        out.append(rec)
    return out


def main():
    ob, tb, hb, genomes, tax = synth.toy_db()
    minidb.write_db(os.path.join(HERE, "toy_db"), ob, tb, hb)
    db = lit.DB.from_images(ob, tb, hb)
    rng = np.random.default_rng(20251001)
    se = synth.sample_reads(rng, genomes, 400, paired=False, len_jitter=110)
    g = genomes[111]
    se += [b"", g[:34], g[:35], b"N" * 80, g[50:200].lower(), g[:40] + b"N" + g[41:150]]
    pe = synth.sample_reads(rng, genomes, 250, paired=True, len_jitter=60)
    pe += [(g[:150], b""), (b"ACGT", g[100:250]), (g[:35], g[35:70])]
    write_fastq(os.path.join(HERE, "reads_se.fq"), se, "se")
    write_fastq(os.path.join(HERE, "reads_pe_1.fq"), [p[0] for p in pe], "pe", 1)
    write_fastq(os.path.join(HERE, "reads_pe_2.fq"), [p[1] for p in pe], "pe", 2)
    for rule, tag in ((1, ""), (0, "_rule0")):
        db.ambiguity_rule = rule
        meta = {"external_ids": db.external, "parent": db.parent, "confidences": CONFS, "ambiguity_rule": rule}
        with open(os.path.join(HERE, "expected_se%s.json" % tag), "w") as f:
            json.dump({"meta": meta, "records": expected(db, se, False)}, f)
        with open(os.path.join(HERE, "expected_pe%s.json" % tag), "w") as f:
            json.dump({"meta": meta, "records": expected(db, pe, True)}, f)
    # known answers of the primitive functions
    kat = {"fmix64": [], "revcomp": []}
    for x in [0, 1, 2, 0xDEADBEEF, (1 << 62) - 1, 0x123456789ABCDEF, 0xFFFFFFFFFFFFFFFF]:
        kat["fmix64"].append([str(x), str(lit.fmix64(x))])
    for n in (1, 15, 31):
        for x in [0, 1, 0x2AAAAAAAAAAAAAAA & ((1 << (2 * n)) - 1), 0x1B1B1B1B1B1B1B1B & ((1 << (2 * n)) - 1)]:
            for rv in (0, 1):
                kat["revcomp"].append([str(x), n, rv, str(lit.reverse_complement(x, n, rv))])
    kat["default_spaced_mask"] = str(minidb.default_spaced_mask())
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f)
    print("fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
