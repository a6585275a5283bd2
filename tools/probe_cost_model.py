"""How many 128-byte lines (and 64-byte sectors) does a kraken2 Get() touch, by table layout?
Linear probing at load 0.7, miss queries (the run ends at the first empty cell), home cells uniform.
Cost model from profiles/r02_pair_study.txt: the chip's random-gather ceiling is a rate of 128-byte
lines; the second sector of a line that is fetched anyway costs ~0.21 of a first one.
    python tools/probe_cost_model.py [load=0.7]"""
import sys

import numpy as np

load = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
rng = np.random.default_rng(1)
cap = 1 << 24
n = int(cap * load)
occ = np.zeros(cap, dtype=bool)
pos = rng.integers(0, cap, n)
pending = np.arange(n)
while pending.size:  # linear-probing inserts, vectorised: the first claimant of a free cell wins
    p = pos[pending]
    order = np.argsort(p, kind="stable")
    ps = p[order]
    first = np.ones(ps.size, bool)
    first[1:] = ps[1:] != ps[:-1]
    win = first & (~occ[ps])
    occ[ps[win]] = True
    pending = pending[order][~win]
    pos[pending] = (pos[pending] + 1) % cap
emp = np.flatnonzero(~occ)
q = rng.integers(0, cap, 4_000_000)
nxt = emp[np.searchsorted(emp, q) % emp.size]
length = ((nxt - q) % cap) + 1  # cells examined, the empty one included
print("load %.2f: mean cells per miss %.2f; P(len > 4, 8, 16, 32) = %s" % (
    load, length.mean(), ", ".join("%.3f" % (length > x).mean() for x in (4, 8, 16, 32))))


def cost(off, name):
    """off = cells between the start of the 128-byte line and the home cell, in the copy used"""
    end = off + length
    lines = (end + 31) // 32
    sectors = (end + 15) // 16 - off // 16
    c = lines + 0.21 * (sectors - lines)
    print("%-44s lines %.4f  sectors %.4f  cost %.4f" % (name, lines.mean(), sectors.mean(), c.mean()))


h = q % 32
cost(h, "1 copy")
cost(np.where(h & 8, (h + 8) % 32, h), "2 copies, 8-cell shift by home&8 (round 1)")
cost(h % 16, "2 copies, 16-cell shift (round 2 default)")
cost(h % 8, "4 copies, 8-cell shift")
cost(h % 4, "8 copies, 4-cell shift")
