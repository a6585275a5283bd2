"""GPU busy time from a rocprofv3 --kernel-trace CSV: union of the dispatches' intervals, per-kernel sums, the window from the first
to the last dispatch of the run's kernels (nh::).   python tools/busy_from_trace.py <kernel_trace.csv> [name filter=nh::]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "nh::"
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if flt in r["Kernel_Name"] and "k_synth" not in r["Kernel_Name"])
if not iv:
    sys.exit("no dispatches")
# the run proper: the last burst of activity (the engine's set-up kernels lie seconds before it)
t_end = iv[-1][1]
start_i = 0
for i in range(len(iv) - 1, 0, -1):
    if iv[i][0] - iv[i - 1][1] > 300_000_000:  # a gap of 0.3 s: what lies before is another run
        start_i = i
        break
iv = iv[start_i:]
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.Counter()
cnt = collections.Counter()
for s, e, k in iv:
    name = k.split("(")[0].replace("void ", "").replace("nh::gz::", "").replace("nh::dfl::", "").replace("nh::fq::", "").replace("nh::", "")[:40]
    per[name] += e - s
    cnt[name] += 1
print("window %.3f s, some kernel running %.3f s = %.1f %% ; sum of kernel durations %.3f s (overlap factor %.2f)" % (
    (t1 - t0) / 1e9, busy / 1e9, 100.0 * busy / (t1 - t0), sum(per.values()) / 1e9, sum(per.values()) / max(busy, 1)))
for k, v in per.most_common(14):
    print("  %-40s %6d dispatches %9.1f ms  %5.1f %%" % (k, cnt[k], v / 1e6, 100.0 * v / sum(per.values())))
# timeline: busy fraction per 50 ms, the kernels that ran in it (ms)
BIN = 50_000_000
nb = int((t1 - t0) / BIN) + 1
marks = [collections.Counter() for _ in range(nb)]
cover = [0] * nb
ev = []
for s, e, k in iv:
    name = k.split("(")[0].replace("void ", "").replace("nh::gz::", "").replace("nh::dfl::", "").replace("nh::fq::", "").replace("nh::", "")[:14]
    b = int((s - t0) / BIN)
    while b < nb and t0 + b * BIN < e:
        lo, hi = max(s, t0 + b * BIN), min(e, t0 + (b + 1) * BIN)
        if hi > lo:
            marks[b][name] += hi - lo
        b += 1
# union per bin
merged = []
cs, ce = iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > ce:
        merged.append((cs, ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
merged.append((cs, ce))
for s, e in merged:
    b = int((s - t0) / BIN)
    while b < nb and t0 + b * BIN < e:
        lo, hi = max(s, t0 + b * BIN), min(e, t0 + (b + 1) * BIN)
        if hi > lo:
            cover[b] += hi - lo
        b += 1
print("timeline, 50 ms a line: busy %, then ms of kernel time by kernel")
for b in range(nb):
    print("  %5.2f s %4.0f %%  %s" % (b * BIN / 1e9, 100.0 * cover[b] / BIN, "  ".join("%s %.0f" % (k, v / 1e6) for k, v in marks[b].most_common(5))))
