#!/bin/bash
# rocprofv3 over one pass of the GPU gzip reader (nh_gunzip.hip): kernel trace + PMC passes (separate runs, as the guide
# prescribes).  usage (on the GPU box): bash scripts/profile_gunzip.sh r04 [extra env assignments for the reader]
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_gunzip
mkdir -p "$OUT"
export TMPDIR=/tmp
GZ=/dev/shm/nh_prof_input.fq.gz
[ -f $GZ ] || python3 tools/gz_make_input.py $GZ 3000000 5 > "$OUT/input.txt" 2>&1
cd /tmp
RUN="python3 $REPO/tools/gz_prof_run.py $GZ"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- $RUN > "$OUT/trace.out" 2> "$OUT/trace.err"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/pmc1" -o pmc1 -- $RUN > /dev/null 2> "$OUT/pmc1.err"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM -d "$OUT/pmc2" -o pmc2 -- $RUN > /dev/null 2> "$OUT/pmc2.err"
rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES_LT_64 -d "$OUT/pmc3" -o pmc3 -- $RUN > /dev/null 2> "$OUT/pmc3.err"
cd "$REPO"
python3 - "$OUT" "$TAG" > "$OUT/summary.txt" <<'PY'
import csv, glob, re, sys, collections, os
out, tag = sys.argv[1], sys.argv[2]
head = os.environ.get("NOHUMAN_GIT_HEAD", "unknown (snapshot without .git)")
tr = open(out + "/trace.out").read()
m = re.search(r"pieces (\d+) chunks (\d+) text (\d+)", tr)
pieces, chunks, text = (int(x) for x in m.groups()) if m else (0, 0, 0)
inp = open(out + "/input.txt").read().strip().splitlines()[-1] if os.path.exists(out + "/input.txt") else ""
inp += " (5 members of 3 000 000 records of 150 bp, Illumina-style ids, binned qualities: 4.30 : 1)"
print("# profiles/%s_inflate_summary.txt -- rocprofv3 over ONE pass of the gzip reader on the GPU (nh_gunzip.hip) on 1 x MI355X" % tag)
print("# command: bash scripts/profile_gunzip.sh %s   (tools/gz_prof_run.py: nh_gunzip_device_file with the reader's defaults -- pieces of 512 MiB" % tag)
print("#          of gzip in chunks of 64 KiB --, text to a device buffer, written to /dev/null; input: tools/gz_make_input.py, bench.py's e2e FASTQ text, level 6)")
print("# source: git HEAD %s" % head)
print("# builder's notes (versions of the kernels, size sweeps, what was tried and dropped): profiles/r04_inflate_notes.txt")
print("# input: %s ; pass: %d pieces, %d chunks, %d bytes of text" % (inp, pieces, chunks, text))
print()
print("== rocprofv3 --kernel-trace --stats, the reader's kernels (ms per pass, GB/s of TEXT = text bytes / kernel time)")
rows = []
for row in csv.reader(open(glob.glob(out + "/trace/*kernel_stats.csv")[0])):
    if row and "nh::" in row[0]:
        rows.append((row[0].split("(")[0].replace("nh::gz::", "").replace("nh::", ""), int(row[1]), float(row[2])))
tot = sum(r[2] for r in rows)
print("  %-22s %6s %10s %8s %10s" % ("kernel", "calls", "total ms", "share", "GB/s text"))
for k, n, ns in rows:
    print("  %-22s %6d %10.2f %7.1f%% %10.1f" % (k, n, ns / 1e6, 100 * ns / tot, text / ns if ns else 0))
print("  %-22s %6s %10.2f %8s %10.1f   <- all kernels of the pass, back to back" % ("sum", "", tot / 1e6, "", text / tot if tot else 0))
for p in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/%s/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("nh::gz::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "inflate" in k or "search" in k or "crc" in k or "scan_local" in k or "resolve" in k:
            print("== %s %s (sum over the pass's launches)" % (p, k))
            for c, v in sorted(d.items()):
                print("  %-24s %.6g" % (c, v))
PY
cat "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete 2>/dev/null
rm -f "$OUT"/trace/*kernel_trace.csv
