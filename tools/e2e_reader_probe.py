"""nh_run on a prepared gzip pair with NOHUMAN_TRACE, both gzip readers, nothing kept / plain / gzip out.
    python tools/e2e_reader_probe.py r_1.fq.gz r_2.fq.gz"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nohuman_amd import Engine
f1, f2 = sys.argv[1], sys.argv[2]
cap = 134_217_689
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
os.environ["NOHUMAN_TRACE"] = "1"
d = os.path.dirname(f1)
for rep in range(2):
    for reader in ("device", "host"):
        for what, kw in (("nothing kept", dict(keep_human=True)), ("gzip out", dict(out_codec=2, codec_threads=8)), ("plain out", dict())):
            os.environ["NOHUMAN_GZ_READER"] = reader
            t = time.perf_counter()
            st = eng.run(f1, os.path.join(d, "o1"), in2=f2, out2=os.path.join(d, "o2"), threads=16, **kw)
            dt = time.perf_counter() - t
            print("RESULT rep %d reader=%s %s: %.3f s = %.1f Mreads/s" % (rep, reader, what, dt, 2 * st.total_sequences / dt / 1e6), flush=True)
            sys.stderr.flush()
