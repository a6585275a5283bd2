"""Throughput and ratio of the GPU gzip encoder on FASTQ text like bench.py's end-to-end inputs, with zlib -6 (what the
reference's gzp runs per block, compression.rs:214-233) timed beside it on one core.
usage: gzip_bench.py [MB of text, default 1024]"""
import ctypes as C
import os, sys, time, types, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nohuman_amd import _lib

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cx = types.SimpleNamespace(torch=torch, dev=torch.device("cuda", 0))
tmp = "/dev/shm/gzip_bench"
os.makedirs(tmp, exist_ok=True)
n_reads = 400_000
parts = []
for m in range(3):
    p = os.path.join(tmp, "m%d.fq" % m)
    bench.e2e_member(cx, n_reads, 150, 1, m, p)
    parts.append(open(p, "rb").read())
unit = b"".join(parts)
data = unit * max(1, (mb << 20) // len(unit))
print("text: %.1f MB (%d distinct MB in rotation)" % (len(data) / 1e6, len(unit) / 1e6), flush=True)
L = _lib.lib()
buf = (C.c_char * len(data)).from_buffer_copy(data)
out = os.path.join(tmp, "out.gz")
for rep in range(3):
    st = (C.c_uint64 * 2)()
    t = time.time()
    rc = L.nh_gzip_gpu_file(0, buf, len(data), os.fsencode(out), st)
    dt = time.time() - t
    assert rc == 0, L.nh_last_error().decode()
    print("GPU gzip: %.3f : 1   wall %.3f s = %.2f GB/s   kernels %.1f ms = %.1f GB/s"
          % (len(data) / st[0], dt, len(data) / dt / 1e9, st[1] / 1e3, len(data) / (st[1] / 1e6) / 1e9), flush=True)
tiny = (C.c_char * 1000).from_buffer_copy(data[:1000])
t = time.time()
for _ in range(5):
    assert L.nh_gzip_gpu_file(0, tiny, 1000, os.fsencode(out + ".tiny"), None) == 0
print("a call on 1000 bytes (buffers allocated, one region, torn down): %.1f ms" % ((time.time() - t) / 5 * 1e3), flush=True)
t = time.time()
ok = zlib.decompress(open(out, "rb").read(), 31) == data
print("zlib inflates it to the text: %s (%.1f s)" % (ok, time.time() - t))
sample = unit[:64 << 20]
t = time.time()
z = zlib.compress(sample, 6)
dt = time.time() - t
print("zlib -6, one core: %.3f : 1   %.1f MB/s" % (len(sample) / len(z), len(sample) / dt / 1e6))
for f in os.listdir(tmp):
    os.unlink(os.path.join(tmp, f))
os.rmdir(tmp)
