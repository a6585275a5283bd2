#!/bin/bash
# one PMC pass of bench.py with a given library:  bash scripts/pmc_one.sh <libname> "<counters>"
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc1_$1; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
NOHUMAN_ENGINE_LIB=$REPO/nohuman_amd/lib$1.so rocprofv3 --output-format csv --pmc $2 -d "$OUT" -o p -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-variants > /dev/null 2> "$OUT.err"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_classify_short" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1].split("_")[-1], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
find "$OUT" -name "*.db" -delete 2>/dev/null
