"""Same-box A / B of nh_run on a prepared gzip pair: every setting (a string of env assignments) is run `reps` times, interleaved.
    python tools/e2e_ab.py r_1.fq.gz r_2.fq.gz plain|gzip|none|se-plain|se-gzip|se-none reps "A=1 B=2" "A=0" ...   ("" = defaults)
Prints wall seconds per run and the process's CPU seconds (user + system) of the median run: how many cores' worth the run kept busy.
NH_AB_DEVICE_FLAGS=n in the CALLER's environment: hipSetDeviceFlags(n) before the engine is opened (4 = hipDeviceScheduleBlockingSync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nohuman_amd import Engine
f1, f2, what, reps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
settings = sys.argv[5:] or [""]
cap = 134_217_689
if os.environ.get("NH_AB_DEVICE_FLAGS"):
    import ctypes
    from nohuman_amd import _lib
    _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(%s) ->" % os.environ["NH_AB_DEVICE_FLAGS"], hip.hipSetDeviceFlags(ctypes.c_uint(int(os.environ["NH_AB_DEVICE_FLAGS"]))), flush=True)
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
d = os.path.dirname(f1)
se = what.startswith("se-")  # single-end: the first file only
what = what[3:] if se else what
kw = {"plain": {}, "gzip": dict(out_codec=2, codec_threads=8), "none": dict(keep_human=True)}[what]
res = {s: [] for s in settings}
cpu = {s: [] for s in settings}
for rep in range(reps + 1):
    for s in settings:
        keys = []
        for kv in s.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
            keys.append(k)
        for o in ("o1", "o2"):
            if os.path.exists(os.path.join(d, o)):
                os.remove(os.path.join(d, o))
        t = time.perf_counter()
        c0 = os.times()
        st = eng.run(f1, os.path.join(d, "o1"), in2=None if se else f2, out2=None if se else os.path.join(d, "o2"), threads=16, **kw)
        dt = time.perf_counter() - t
        c1 = os.times()
        for k in keys:
            os.environ.pop(k, None)
        if rep:  # (the first round warms buffers and page cache)
            res[s].append(dt)
            cpu[s].append((c1.user - c0.user) + (c1.system - c0.system))
for s in settings:
    v = sorted(res[s])
    c = sorted(cpu[s])[len(cpu[s]) // 2]
    print("%-40s %s  median %.3f s = %.1f Mreads/s; CPU %.1f s = %.1f cores" % (s or "(defaults)", " ".join("%.3f" % x for x in res[s]), v[len(v) // 2],
                                                                        (1 if se else 2) * st.total_sequences / v[len(v) // 2] / 1e6, c, c / v[len(v) // 2]), flush=True)
