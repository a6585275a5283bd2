#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (scripts/profile.sh) into the committed summary
profiles/<round>_summary.txt, profiles/<round>_kernel_stats.csv and profiles/traffic.json.
    python scripts/make_profile_summary.py gpurun_out/prof_r01c r01"""
import csv, glob, json, os, sys
src, rnd = sys.argv[1], sys.argv[2]
KERNEL = 'k_classify_short'  # the kernel the bench workload (150 bp pairs) runs in; the generic k_classify behind it returns at once
LOOKUPS = 77.43e6 * 2.5  # 2.5 M pairs per launch


def mean_counter(pattern, kernel_sub, counter):
    vals = []
    for f in glob.glob(os.path.join(src, pattern)):
        for row in csv.DictReader(open(f)):
            if kernel_sub in row['Kernel_Name'] and row['Counter_Name'] == counter:
                vals.append(float(row['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out = []
out.append("# profiles/%s_summary.txt -- rocprofv3 summaries of bench.py on 1 x MI355X" % rnd)
out.append("# command: bash scripts/profile.sh <tag>   (trace pass: bench.py --no-cpu-baseline --no-e2e --no-variants, 20 + 3 launches;")
out.append("#          PMC passes: the same with --steps 5 --warmup 2)")
try:
    bj = json.loads(open(os.path.join(src, 'bench_trace.json')).read().strip().splitlines()[-1])
    out.append("# bench.py's own line in the trace pass: value %.1f %s, ms_per_step %.4f, roofline.kernel_ms %.4f" % (
        bj['value'], bj['unit'], bj['ms_per_step'], bj['roofline']['kernel_ms']))
except Exception as e:
    out.append("# (bench line of the trace pass not available: %s)" % e)
out.append("# workload: 2,500,000 fragments = 5,000,000 x 150 bp PE reads per launch, synthetic HPRC.r2-like table")
out.append("#           1,431,655,765 cells (5.73 GB), load 0.70, k=35 l=31")
out.append("")
out.append("== rocprofv3 --kernel-trace --stats (trace_kernel_stats.csv), our kernels")
for row in csv.reader(open(os.path.join(src, 'trace/trace_kernel_stats.csv'))):
    if row and (row[0] == 'Name' or 'nh::' in row[0]):
        out.append("  " + ", ".join(c[:70] for c in row))
out.append("")
out.append("== PMC passes (separate runs), %s, mean per launch" % KERNEL)
for pat, cs in (('pmc1/*counter_collection.csv', ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES']),
                ('pmc2/*counter_collection.csv', ['SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS']),
                ('pmc3/*counter_collection.csv', ['FETCH_SIZE']),
                ('pmc4/*counter_collection.csv', ['TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_HIT_sum', 'TCC_MISS_sum']),
                ('pmc5/*counter_collection.csv', ['WRITE_SIZE'])):
    for c in cs:
        v, n = mean_counter(pat, KERNEL, c)
        if v is not None:
            out.append("  %-26s %.6g   (n=%d)" % (c, v, n))
fs, _ = mean_counter('pmc3/*counter_collection.csv', KERNEL, 'FETCH_SIZE')
ws, _ = mean_counter('pmc5/*counter_collection.csv', KERNEL, 'WRITE_SIZE')
rq, _ = mean_counter('pmc4/*counter_collection.csv', KERNEL, 'TCC_EA0_RDREQ_sum')
cf, _ = mean_counter('calib/*counter_collection.csv', 'k_gather_mode<0>', 'FETCH_SIZE')
cr, _ = mean_counter('calib/*counter_collection.csv', 'k_gather_mode<0>', 'TCC_EA0_RDREQ_sum')
if cf is not None:
    probes = 8192 * 64 * 256
    out.append("")
    out.append("== FETCH_SIZE calibration on a known pattern (tools/gather_bench under rocprofv3, same run)")
    out.append("  k_gather_mode<0>: %d random 16-byte probes per launch -> TCC_EA0_RDREQ_sum %.6g (%.3f per probe)," % (probes, cr, cr / probes))
    out.append("  FETCH_SIZE %.6g KB (%.1f B per probe): on a random gather FETCH_SIZE = 64 B x fabric read requests," % (cf, cf * 1024 / probes))
    out.append("  one request per missing line.  The chip sustains ~52e9 such requests/s whatever the cache policy bits")
    out.append("  (profiles/%s_gather_bench.txt)." % rnd)
out.append("")
out.append("== derived, per launch")
out.append("  lookups D                   %.4g   (38.7 per read)" % LOOKUPS)
out.append("  fabric read requests        %.4g   (%.2f per lookup, incl. ~1.2e7 for the streamed bases)" % (rq, rq / LOOKUPS))
out.append("  FETCH_SIZE as counted       %.4g bytes ; WRITE_SIZE %.4g bytes" % (fs * 1024, ws * 1024))
out.append("  HBM bytes (128 B/request)   %.4g   = 2 x FETCH_SIZE + WRITE_SIZE (every fabric read is a 128-byte request: r02_mem_study.txt)" % (2 * fs * 1024 + ws * 1024))
out.append("  algorithmic bytes           1.318e10  (sum len + 64*D + 16 per fragment, BASELINE.md section 4)")
os.makedirs('profiles', exist_ok=True)
open('profiles/%s_summary.txt' % rnd, 'w').write("\n".join(out) + "\n")
with open('profiles/%s_kernel_stats.csv' % rnd, 'w') as g:
    for row in csv.reader(open(os.path.join(src, 'trace/trace_kernel_stats.csv'))):
        g.write(",".join('"%s"' % c[:120] for c in row) + "\n")
json.dump({"workload": {"fragments_per_step": 2500000, "paired": True, "read_len": 150, "capacity": 1431655765},
           "source": "profiles/%s_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, k_classify_short mean per launch)" % rnd,
           "fetch_size_kb": fs, "write_size_kb": ws,
           "note": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE: the guide's gfx950 correction (FETCH_SIZE tallies 128-byte requests "
                   "at 64 B), confirmed for THIS access pattern by profiles/r02_mem_study.txt: TCC_EA0_RDREQ_128B_sum == "
                   "TCC_EA0_RDREQ_sum, every fabric read of the kernel (and of the random-gather microbenchmark) is a "
                   "128-byte request",
           "traffic_bytes_per_launch": int(2 * fs * 1024 + ws * 1024),
           "fetch_size_as_counted_bytes": int(fs * 1024),
           "fabric_read_requests_per_launch": int(rq),
           "fabric_request_ceiling_per_s": 50e9,
           "hbm_achievable_gbs": 6290.0,
           "ceiling_source": "tools/gather_bench (profiles/%s_gather_bench.txt): a pure random gather sustains ~50e9 "
                             "128-byte fabric reads/s = 6.4 TB/s, the HBM bandwidth the chip reaches on a copy "
                             "(6.29 TB/s, MI355X_MICROARCH.md)" % rnd},
          open('profiles/traffic.json', 'w'), indent=1)
print("\n".join(out))
