"""nh_gunzip_file at several chunk sizes and worker counts on one realistic member (bench.py's e2e generator):
wall and CPU seconds.  usage: gunzip_chunks.py [records=3000000]"""
import ctypes as C, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nohuman_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
class Cx: pass
cx = Cx(); cx.torch = torch; cx.np = np
cx.dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
tmp = tempfile.mkdtemp(prefix="gzc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
plain = os.path.join(tmp, "m.fq")
size = bench.e2e_member(cx, n, 150, 1, 3, plain)
gz = plain + ".gz"
assert L.nh_compress_file(os.fsencode(plain), os.fsencode(gz), 2, 16) == 0
print("%d records, %.2f GB of text, %.2f GB gzip -6" % (n, size / 1e9, os.path.getsize(gz) / 1e9), flush=True)
for thr in (8, 16):
    for chunk in (256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20):
        w = []; c = []
        for rep in range(3):
            c0 = time.process_time(); t0 = time.perf_counter()
            assert L.nh_gunzip_file(os.fsencode(gz), b"/dev/null", thr, chunk, None) == 0
            w.append(time.perf_counter() - t0); c.append(time.process_time() - c0)
        print("%2d workers, chunks of %5d KiB: wall %.3f s = %.2f GB/s of text; CPU %.2f s = %.2f GB/s per core-second"
              % (thr, chunk >> 10, min(w), size / min(w) / 1e9, min(c), size / min(c) / 1e9), flush=True)
shutil.rmtree(tmp)
