"""Where does a gzip FASTQ input's time go on this host?  One realistic member (bench.py's e2e generator), then
nh_gunzip_file (decoder alone, output to /dev/null) and nh_fastx_scan (decoder + record parser) at several worker
counts: wall and CPU seconds.  usage: gunzip_scaling.py [records=2000000]"""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nohuman_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
class Cx: pass
cx = Cx(); cx.torch = torch; cx.np = np
cx.dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="gzs_", dir=base)
plain = os.path.join(tmp, "m.fq")
size = bench.e2e_member(cx, n, 150, 1, 3, plain)
gz = plain + ".gz"
t = time.perf_counter(); assert L.nh_compress_file(os.fsencode(plain), os.fsencode(gz), 2, 16) == 0
print("%d records, %.2f GB of text -> %.2f GB gzip -6 in %.1f s (16 encoder threads)" % (n, size / 1e9, os.path.getsize(gz) / 1e9, time.perf_counter() - t))
for thr in (1, 2, 4, 8, 16):
    w = []; c = []
    for rep in range(3):
        c0 = time.process_time(); t0 = time.perf_counter()
        assert L.nh_gunzip_file(os.fsencode(gz), b"/dev/null", thr, 0, None) == 0
        w.append(time.perf_counter() - t0); c.append(time.process_time() - c0)
    print("decoder alone, %2d workers: wall %.3f s = %.2f GB/s of text; CPU %.2f s = %.2f GB/s per core-second" % (thr, min(w), size / min(w) / 1e9, min(c), size / min(c) / 1e9))
for thr in (1, 4, 8, 16):
    os.environ["NOHUMAN_GZ_THREADS"] = str(thr)
    w = []; c = []
    for rep in range(3):
        nr, nb, dg = C.c_uint64(), C.c_uint64(), C.c_uint64()
        c0 = time.process_time(); t0 = time.perf_counter()
        assert L.nh_fastx_scan(os.fsencode(gz), C.byref(nr), C.byref(nb), C.byref(dg)) == 0
        w.append(time.perf_counter() - t0); c.append(time.process_time() - c0)
    print("decoder + parser + digest (nh_fastx_scan), %2d workers: wall %.3f s = %.2f GB/s; CPU %.2f s" % (thr, min(w), size / min(w) / 1e9, min(c)))
w = []
for rep in range(3):
    nr, nb, dg = C.c_uint64(), C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter(); assert L.nh_fastx_scan(os.fsencode(plain), C.byref(nr), C.byref(nb), C.byref(dg)) == 0; w.append(time.perf_counter() - t0)
print("plain text, parser + digest: wall %.3f s = %.2f GB/s" % (min(w), size / min(w) / 1e9))
import shutil; shutil.rmtree(tmp)
