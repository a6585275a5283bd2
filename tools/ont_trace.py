"""Stage trace (NOHUMAN_TRACE) of nh_run on bench.py's ONT-shaped input: ONE gzip file of long reads, gzip -> gzip and the input side alone.
    python tools/ont_trace.py [reads per member=100000] [members=5]        (settings: env, e.g. NOHUMAN_GZ_READER=host)"""
import os, shutil, sys, tempfile, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from nohuman_amd import Engine
n_member = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cx = types.SimpleNamespace(torch=torch, np=np, dev=torch.device("cuda", 0))
cap = 134_217_689
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
tmp = tempfile.mkdtemp(prefix="nh_ont_trace_", dir="/dev/shm")
try:
    fin, text_len, _, n50 = bench.make_ont_input(cx, tmp, n_member, members, 16)
    print("input: %.2f GB of gzip, %.2f GB of text, N50 %d" % (os.path.getsize(fin) / 1e9, text_len * members / 1e9, n50), flush=True)
    out = os.path.join(tmp, "o.fq.gz")
    for what, kw in (("gzip -> gzip", dict(out_codec=2, codec_threads=8)), ("input side alone", dict(keep_human=True))):
        for rep in range(3):
            if os.path.exists(out):
                os.remove(out)
            if rep == 2:
                os.environ["NOHUMAN_TRACE"] = "1"
                sys.stderr.write("==== %s, traced run\n" % what)
                sys.stderr.flush()
            t = time.perf_counter()
            st = eng.run(fin, out, threads=16, **kw)
            dt = time.perf_counter() - t
            os.environ.pop("NOHUMAN_TRACE", None)
            print("%-18s run %d: %.3f s = %.3f Mreads/s = %.2f Gbases/s" % (what, rep, dt, st.total_sequences / dt / 1e6, st.total_bases / dt / 1e9), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
