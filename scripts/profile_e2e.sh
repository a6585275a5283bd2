#!/bin/bash
# Kernel-trace statistics of the whole files-in -> files-out leg of bench.py (plain and gzip outputs): which kernels
# the GPU spends its time in during nh_run.   gpurun -- 'bash scripts/profile_e2e.sh r03'
set -u
TAG=${1:-r03}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_${TAG}_e2e
mkdir -p "$OUT"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $REPO/bench.py --no-variants --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/trace.err"
cd "$REPO"
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh "$OUT"
