#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel stats CSV + counter_collection CSVs) into a
small text summary that can be committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                print("  " + ", ".join(c[:90] for c in row))
for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, out))
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")[:60]
            acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    for k, cs in acc.items():
        print("  kernel:", k)
        for c, v in sorted(cs.items()):
            print("    %-28s n=%-4d mean=%.6g  last=%.6g" % (c, len(v), sum(v) / len(v), v[-1]))
