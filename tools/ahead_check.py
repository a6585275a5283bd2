"""Full-size check of the reader over several lanes (pieces decoded ahead): nh_run on a prepared gzip pair with the toy database,
one engine against several engines on GPU 0: output digests, counters, wall time, the reader's trace lines.
    python tools/ahead_check.py r_1.fq.gz r_2.fq.gz [lanes=2]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nohuman_amd import engine
f1, f2 = sys.argv[1], sys.argv[2]
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
DB = os.path.join(ROOT, "tests", "golden", "toy_db")
d = os.path.dirname(f1)
os.environ["NOHUMAN_TRACE"] = "1"
res = {}
for tag, ids in (("one", [0]), ("lanes", [0] * lanes), ("lanes_again", [0] * lanes)):
    o1, o2 = os.path.join(d, "o1"), os.path.join(d, "o2")
    for o in (o1, o2):
        if os.path.exists(o):
            os.remove(o)
    t = time.perf_counter()
    st = engine.run(DB, f1, o1, in2=f2, out2=o2, device_ids=ids, threads=16)
    dt = time.perf_counter() - t
    md = subprocess.run(["md5sum", o1, o2], capture_output=True, text=True).stdout.split()
    res[tag] = (md[0], md[2], st.total_sequences, st.classified, st.total_bases)
    print("%-12s %.3f s  %s" % (tag, dt, res[tag]), flush=True)
print("EQUAL" if res["one"] == res["lanes"] == res["lanes_again"] else "DIFFERENT")
