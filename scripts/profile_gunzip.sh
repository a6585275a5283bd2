#!/bin/bash
# rocprofv3 over one pass of the GPU gzip reader (nh_gunzip.hip): kernel trace + PMC passes (separate runs, as the guide
# prescribes).  usage (on the GPU box): bash scripts/profile_gunzip.sh r04 [extra env assignments for the reader]
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_gunzip
mkdir -p "$OUT"
export TMPDIR=/tmp
GZ=/dev/shm/nh_prof_input.fq.gz
[ -f $GZ ] || python3 tools/gz_make_input.py $GZ 3000000 2 > "$OUT/input.txt" 2>&1
cd /tmp
RUN="python3 $REPO/tools/gz_prof_run.py $GZ"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- $RUN > "$OUT/trace.out" 2> "$OUT/trace.err"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc1" -o pmc1 -- $RUN > /dev/null 2> "$OUT/pmc1.err"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM -d "$OUT/pmc2" -o pmc2 -- $RUN > /dev/null 2> "$OUT/pmc2.err"
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES_LT_64 -d "$OUT/pmc3" -o pmc3 -- $RUN > /dev/null 2> "$OUT/pmc3.err"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
print("== kernel stats")
for row in csv.reader(open(glob.glob(out + "/trace/*kernel_stats.csv")[0])):
    if row and (row[0] == "Name" or "nh::" in row[0]):
        print("  " + ", ".join(c[:60] for c in row[:8]))
for p in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/%s/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-24:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "inflate" in k or "search" in k:
            print("== %s %s (sum over launches)" % (p, k))
            for c, v in sorted(d.items()):
                print("  %-24s %.6g" % (c, v))
PY
find "$OUT" -name "*.db" -delete 2>/dev/null
rm -f "$OUT"/trace/*kernel_trace.csv
