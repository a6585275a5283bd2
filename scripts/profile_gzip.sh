#!/bin/bash
# PMC profile of the GPU gzip encoder's kernel (k_deflate) on tools/gzip_bench.py's text.
#   gpurun -- 'bash scripts/profile_gzip.sh r03'
set -u
TAG=${1:-r03}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_${TAG}_gzip
mkdir -p "$OUT"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $REPO/tools/gzip_bench.py 512 > "$OUT/bench_trace.txt" 2> "$OUT/trace.err"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d "$OUT/pmc1" -o pmc1 -- python3 $REPO/tools/gzip_bench.py 256 > /dev/null 2> "$OUT/pmc1.err"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d "$OUT/pmc2" -o pmc2 -- python3 $REPO/tools/gzip_bench.py 256 > /dev/null 2> "$OUT/pmc2.err"
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM -d "$OUT/pmc3" -o pmc3 -- python3 $REPO/tools/gzip_bench.py 256 > /dev/null 2> "$OUT/pmc3.err"
# the texture / L1 path (what the match finder's gathers go through): is it the address unit that is busy?
rocprofv3 --output-format csv --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d "$OUT/pmc4" -o pmc4 -- python3 $REPO/tools/gzip_bench.py 256 > /dev/null 2> "$OUT/pmc4.err"
rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d "$OUT/pmc5" -o pmc5 -- python3 $REPO/tools/gzip_bench.py 256 > /dev/null 2> "$OUT/pmc5.err"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "deflate" in r.get("Name", ""):
            print("stats", r.get("Name")[:60], "calls", r.get("Calls"), "avg ns", r.get("AverageNs"), "total ns", r.get("TotalDurationNs"))
for d in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob(out + "/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "k_deflateILi" in r["Kernel_Name"] or "k_deflate<" in r["Kernel_Name"]:  # (mangled or demangled, by rocprofv3 version)
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n += 1
    disp = n / max(1, len(acc))
    for k, v in sorted(acc.items()):
        print("%s %-24s %.4g per dispatch (%d dispatches)" % (d, k, v / max(1, disp), disp))
PY
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh "$OUT"
