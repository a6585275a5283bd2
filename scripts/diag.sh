#!/bin/bash
# quick diagnosis of k_classify on the bench workload: phase profile + two PMC passes
#   gpurun -- 'bash scripts/diag.sh tag'
TAG=${1:-d}; REPO=$(pwd); OUT=$REPO/gpurun_out/diag_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 tools/phase_prof.py > "$OUT/phase.txt" 2>&1
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-variants"
cd /tmp
rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/p1" -o p1 -- $BENCH > "$OUT/p1.json" 2> "$OUT/p1.err"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d "$OUT/p2" -o p2 -- $BENCH > /dev/null 2> "$OUT/p2.err"
cd "$REPO"
cat "$OUT/phase.txt"
python3 scripts/summarize_prof.py "$OUT" | grep -A12 "k_classify" | cut -c1-110
find "$OUT" -name "*.db" -delete 2>/dev/null
