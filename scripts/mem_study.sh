#!/bin/bash
# memory-side counters of k_classify_short next to the pure gather microbenchmark.
# NOTE: tools/gather_bench under rocprofv3 --pmc is SLOW (minutes per pass: hundreds of launches with counters); the
# first two counter sets took 30 GPU-minutes in round 2.  Run single sets with scripts/pmc_one.sh when in doubt.
REPO=$(pwd); OUT=$REPO/gpurun_out/mem; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-variants"
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum TCC_BUSY_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_NORMAL_EVICT_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d "$OUT/k$i" -o p -- $BENCH > /dev/null 2> "$OUT/k$i.err"
  rocprofv3 --output-format csv --pmc $set -d "$OUT/g$i" -o p -- "$REPO/tools/gather_bench" 6 64 > /dev/null 2> "$OUT/g$i.err"
done
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
for tag in ("k", "g"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/mem/%s*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "k_classify_short" in kn or "k_gather1" in kn:
                agg[(kn[:40], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-42s %-40s grid=%-9s n=%-3d mean=%.5g" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
find "$OUT" -name "*.db" -delete 2>/dev/null
