#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats, then PMC passes (separate runs, as the
# MI355X guide prescribes: no --pmc together with trace domains other than --kernel-trace).
#   gpurun -- 'bash scripts/profile.sh r03 pe se hit ont'     (workloads: pe = the bench default, se = configs[1],
#                                                              hit = hit fraction 0.5, ont = configs[3] shape, wide,
#                                                              pe250 = 2 x 250 bp pairs, n = pairs with 0.1 % N)
set -u
TAG=${1:-r03}
shift
WORKLOADS=${*:-pe}
REPO=$(pwd)
export TMPDIR=/tmp
# the GPU box runs a snapshot without .git: the caller passes the commit (NOHUMAN_GIT_HEAD=$(git rev-parse HEAD) in the
# gpurun command line); make_profile_summary.py writes it and a hash of the kernel sources into every summary and into
# profiles/traffic.json, and bench.py reports "traffic_stale": true when the sources have changed since
export NOHUMAN_GIT_HEAD=${NOHUMAN_GIT_HEAD:-$(git rev-parse HEAD 2>/dev/null || true)}
for W in $WORKLOADS; do
  case $W in
    pe)   WARGS="" ; STEPS=20 ;;
    se)   WARGS="--single-end --pairs 1000000" ; STEPS=20 ;;
    hit)  WARGS="--hit-frac 0.5 --pairs 1000000" ; STEPS=20 ;;
    ont)  WARGS="--ont --pairs 200000" ; STEPS=8 ;;
    wide) WARGS="--pairs 1000000 --capacity 4400000011" ; STEPS=10 ;;
    pe250) WARGS="--read-len 250 --pairs 600000" ; STEPS=12 ;;
    n)    WARGS="--n-rate 0.001 --pairs 1000000" ; STEPS=20 ;;
    *) echo "unknown workload $W"; continue ;;
  esac
  OUT=$REPO/gpurun_out/prof_${TAG}_$W
  mkdir -p "$OUT"
  COMMON="--no-cpu-baseline --no-e2e --no-variants $WARGS"
  BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --wake-ms 0 $COMMON"
  cd /tmp
  # the trace pass runs the step count bench.py reports on (3 warm-up launches), so that the
  # average duration of the classify kernel is the steady-state one
  rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $REPO/bench.py --steps $STEPS --warmup 3 $COMMON > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
  rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d "$OUT/pmc1" -o pmc1 -- $BENCH > /dev/null 2> "$OUT/pmc1.err"
  rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d "$OUT/pmc2" -o pmc2 -- $BENCH > /dev/null 2> "$OUT/pmc2.err"
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc3" -o pmc3 -- $BENCH > /dev/null 2> "$OUT/pmc3.err"
  rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d "$OUT/pmc4" -o pmc4 -- $BENCH > /dev/null 2> "$OUT/pmc4.err"
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc5" -o pmc5 -- $BENCH > /dev/null 2> "$OUT/pmc5.err"
  # calibration of FETCH_SIZE on a known pattern: tools/gather_bench issues a known number of random
  # 16-byte probes, each of which misses to one 128-byte line
  if [ "$W" = pe ] && [ -x "$REPO/tools/gather_bench" ]; then
    rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d "$OUT/calib" -o calib -- "$REPO/tools/gather_bench" 6 64 > "$OUT/calib_stdout.txt" 2> "$OUT/calib.err"
  fi
  cd "$REPO"
  python3 scripts/make_profile_summary.py "$OUT" "$TAG" "$W" > "$OUT/summary.txt" 2>&1
  cat "$OUT/summary.txt"
  find "$OUT" -name "*.db" -delete 2>/dev/null
  # the per-dispatch trace is large: keep the stats and the classify kernels' rows only
  for f in "$OUT"/trace/*kernel_trace.csv; do
    [ -f "$f" ] && { head -1 "$f"; grep -E "k_classify|k_prep_items|k_fold_counters" "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
  done
  du -sh "$OUT"
done
