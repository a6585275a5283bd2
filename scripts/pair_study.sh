#!/bin/bash
# Fabric study: is the random-gather ceiling a rate of 64-B sector requests or of
# 128-byte lines?  gpurun -- 'bash scripts/pair_study.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/pair; mkdir -p "$OUT"; export TMPDIR=/tmp
"$REPO/tools/gather_bench" 6 64 1 > "$OUT/pair.txt" 2>&1
"$REPO/tools/valu_bench" > "$OUT/valu.txt" 2>&1
cd /tmp
rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_HIT_sum FETCH_SIZE -d "$OUT/pmc" -o pmc -- "$REPO/tools/gather_bench" 6 64 1 > /dev/null 2> "$OUT/pmc.err"
cd "$REPO"
find "$OUT" -name "*.db" -delete 2>/dev/null
cat "$OUT/pair.txt"; cat "$OUT/valu.txt"
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pair/pmc/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, len(v), sum(v)/len(v))
PY
