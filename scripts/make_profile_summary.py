#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>_<workload>/ directory (scripts/profile.sh) into the committed records:
profiles/<tag>_<workload>_summary.txt, profiles/<tag>_<workload>_kernel_stats.csv and the workload's entry of
profiles/traffic.json (which bench.py reads for `roofline.traffic` of the headline AND of its variants).
    python scripts/make_profile_summary.py gpurun_out/prof_r03_pe r03 pe"""
import csv, glob, hashlib, json, os, subprocess, sys
src, rnd = sys.argv[1], sys.argv[2]


def source_id():
    """What the profiled binary was built from: git HEAD where there is a work tree (the GPU box gets a
    snapshot without .git: scripts/profile.sh passes NOHUMAN_GIT_HEAD when it knows it) and a hash of the kernel sources,
    which bench.py recomputes to say whether profiles/traffic.json still belongs to the code it runs."""
    head = os.environ.get("NOHUMAN_GIT_HEAD", "")
    if not head:
        try:
            head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        except Exception:
            head = ""
    h = hashlib.sha256()
    for f in ("nohuman_amd/csrc/nh_kernels.hip", "nohuman_amd/csrc/nh_device.h"):
        h.update(open(f, "rb").read())
    return head or "unknown (snapshot without .git)", h.hexdigest()[:16]


GIT_HEAD, KERNEL_HASH = source_id()
wl = sys.argv[3] if len(sys.argv) > 3 else "pe"
# the kernel the workload's fragments are classified in (the other one, launched with it, returns at once)
GENERIC = wl in ("ont", "pe250")  # reads longer than one tile: every chunk is left to the generic kernel
KERNEL = 'k_classify<' if GENERIC else 'k_classify_short'
WARMUP = 3  # launches of the trace pass before bench.py's timed region


def mean_counter(pattern, kernel_sub, counter):
    vals = []
    for f in glob.glob(os.path.join(src, pattern)):
        for row in csv.DictReader(open(f)):
            if kernel_sub in row['Kernel_Name'] and row['Counter_Name'] == counter:
                v = float(row['Counter_Value'])
                if GENERIC and counter == 'SQ_WAVES' and v < 100:
                    continue
                vals.append(v)
    # (the generic kernel is also launched as the BIG second pass and, for short reads, as the deferred pass:
    #  those dispatches do nothing; keep the ones that did the work = the larger half by value)
    if GENERIC and vals:
        big = max(vals)
        vals = [v for v in vals if v > 0.2 * big]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out = []
bj = None
try:
    bj = json.loads(open(os.path.join(src, 'bench_trace.json')).read().strip().splitlines()[-1])
except Exception as e:
    out.append("# (bench line of the trace pass not available: %s)" % e)
out.append("# profiles/%s_%s_summary.txt -- rocprofv3 summaries of bench.py on 1 x MI355X, workload '%s'" % (rnd, wl, wl))
out.append("# command: bash scripts/profile.sh %s %s   (trace pass: bench.py with its reported step count behind the wake launches and %d warm-up launches;" % (rnd, wl, WARMUP))
out.append("#          PMC passes: the same with --steps 5 --warmup 2, one rocprofv3 run per counter group)")
out.append("# source: git HEAD %s, sha256(nh_kernels.hip + nh_device.h)[:16] = %s" % (GIT_HEAD, KERNEL_HASH))
if bj:
    out.append("# bench.py's own line in the trace pass: value %.1f %s, ms_per_step %.4f, roofline.kernel_ms %.4f, frac %.4f" % (
        bj['value'], bj['unit'], bj['ms_per_step'], bj['roofline']['kernel_ms'], bj['roofline']['frac']))
    out.append("# workload: " + bj['config']['workload'])
out.append("")
out.append("== rocprofv3 --kernel-trace --stats (trace_kernel_stats.csv), our kernels (ALL dispatches: wake launches, %d warm-ups, the timed steps, the two-stream repetition)" % WARMUP)
for row in csv.reader(open(os.path.join(src, 'trace/trace_kernel_stats.csv'))):
    if row and (row[0] == 'Name' or 'nh::' in row[0]):
        out.append("  " + ", ".join(c[:70] for c in row))
# per-dispatch durations of the classify kernel, warm-ups dropped (the stats mean includes them)
durs = []
for f in glob.glob(os.path.join(src, 'trace/*kernel_trace.csv')):
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
    if GENERIC and d:
        d = [x for x in d if x > 0.2 * max(d)]
    # the trace holds: bench.py's untimed wake launches (roofline.wake_launches), the warm-up steps, the K timed steps, then the
    # two-stream repetition (2 + K launches): the timed region is what counts
    wake = int(bj['roofline'].get('wake_launches', 0)) if bj else 0
    durs = d[wake + WARMUP:wake + WARMUP + int(bj['steps'])] if bj else d[WARMUP:]
steady = None
if durs:
    steady = sum(durs) / len(durs)
    out.append("")
    out.append("== %s, the %d dispatches of the timed region (wake launches and warm-ups dropped): mean %.1f us, min %.1f, max %.1f" % (
        KERNEL.rstrip('<'), len(durs), steady / 1e3, min(durs) / 1e3, max(durs) / 1e3))
    if bj:
        ab = bj['roofline']['algorithmic_bytes_per_launch']
        out.append("   algorithmic bytes per launch %.4g / that mean = %.1f GB/s = %.4f of 8 TB/s (bench.py's own events: %.4f)" % (
            ab, ab / steady, ab / steady / 8000.0, bj['roofline']['frac']))
out.append("")
out.append("== PMC passes (separate runs), %s, mean per launch" % KERNEL.rstrip('<'))
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
    out.append("  one request per missing line.")
entry = None
if fs is not None and ws is not None and rq is not None and bj:
    frags = bj['config']['fragments_per_step']
    mates = 2 if bj['config']['paired'] else 1
    lookups = bj['config']['lookups_per_read'] * frags * mates
    traffic = 2 * fs * 1024 + ws * 1024
    ab = bj['roofline']['algorithmic_bytes_per_launch']
    out.append("")
    out.append("== derived, per launch")
    out.append("  lookups D                   %.4g   (%.1f per read)" % (lookups, bj['config']['lookups_per_read']))
    out.append("  fabric read requests        %.4g   (%.2f per lookup, incl. the streamed bases)" % (rq, rq / lookups))
    out.append("  FETCH_SIZE as counted       %.4g bytes ; WRITE_SIZE %.4g bytes" % (fs * 1024, ws * 1024))
    out.append("  HBM bytes (128 B/request)   %.4g   = 2 x FETCH_SIZE + WRITE_SIZE (every fabric read is a 128-byte request: r02_mem_study.txt)" % traffic)
    out.append("  algorithmic bytes           %.4g   (sum len + 64*D + 16 per fragment, BASELINE.md section 4): traffic / algorithmic = %.2f" % (ab, traffic / ab))
    if steady:
        out.append("  HBM bytes / steady kernel time = %.0f GB/s = %.3f of the 6290 GB/s a copy reaches" % (traffic / steady, traffic / steady / 6290.0))
    entry = {
        "workload": {"fragments_per_step": frags, "paired": bj['config']['paired'], "what": bj['config']['workload']},
        "source": "profiles/%s_%s_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / TCC_EA0_RDREQ, separate passes, %s mean per launch)" % (rnd, wl, KERNEL.rstrip('<')),
        "fetch_size_kb": fs, "write_size_kb": ws,
        "traffic_bytes_per_launch": int(traffic),
        "fetch_size_as_counted_bytes": int(fs * 1024),
        "fabric_read_requests_per_launch": int(rq),
        "algorithmic_bytes_per_launch": int(ab),
        "kernel_us_steady_rocprof": (steady / 1e3) if steady else None,
        "git_head": GIT_HEAD, "kernel_source_sha16": KERNEL_HASH,
    }
os.makedirs('profiles', exist_ok=True)
open('profiles/%s_%s_summary.txt' % (rnd, wl), 'w').write("\n".join(out) + "\n")
with open('profiles/%s_%s_kernel_stats.csv' % (rnd, wl), 'w') as g:
    for row in csv.reader(open(os.path.join(src, 'trace/trace_kernel_stats.csv'))):
        g.write(",".join('"%s"' % c[:120] for c in row) + "\n")
if entry:
    path = 'profiles/traffic.json'
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        t = {}
    if "workloads" not in t:
        t = {"note": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE: the guide's gfx950 correction (FETCH_SIZE tallies 128-byte requests "
                     "at 64 B), confirmed for THIS access pattern by profiles/r02_mem_study.txt: TCC_EA0_RDREQ_128B_sum == "
                     "TCC_EA0_RDREQ_sum, every fabric read of the kernel (and of the random-gather microbenchmark) is a 128-byte request",
             "fabric_request_ceiling_per_s": 50e9, "hbm_achievable_gbs": 6290.0,
             "ceiling_source": "tools/gather_bench (profiles/r02_gather_bench.txt): a pure random gather sustains ~50e9 128-byte fabric "
                               "reads/s = 6.4 TB/s, the HBM bandwidth the chip reaches on a copy (6.29 TB/s, MI355X_MICROARCH.md)",
             "workloads": {}}
    t["workloads"][wl] = entry
    json.dump(t, open(path, 'w'), indent=1)
print("\n".join(out))
