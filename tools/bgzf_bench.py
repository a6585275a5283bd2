"""The gzip reader on the GPU on a BGZF (bgzip) file of bench.py's FASTQ text against the same text as ordinary gzip:
    python tools/bgzf_bench.py [records=3000000]"""
import ctypes as C, os, struct, subprocess, sys, time, types, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from nohuman_amd import _lib
from multiprocessing import Pool
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
cx = types.SimpleNamespace(torch=torch, dev=torch.device("cuda", 0))
plain = "/dev/shm/bgzf_bench.fq"
bench.e2e_member(cx, n, 150, 1, 0, plain)
data = open(plain, "rb").read()


def member(blk):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(blk) + co.flush()
    return (b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body +
            struct.pack("<II", zlib.crc32(blk), len(blk)))


with Pool(8) as pool:
    parts = pool.map(member, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=64)
bg = "/dev/shm/bgzf_bench.fq.bgz.gz"
open(bg, "wb").write(b"".join(parts) + member(b""))
L = _lib.lib()
gzp = "/dev/shm/bgzf_bench.fq.gz"
assert L.nh_compress_file(os.fsencode(plain), os.fsencode(gzp), 2, 16) == 0
code = ("import sys, time, ctypes as C; sys.path.insert(0, %r)\n"
        "from nohuman_amd import _lib\n"
        "L = _lib.lib(); st = (C.c_uint64 * 8)()\n"
        "for rep in range(2):\n"
        "    rc = L.nh_gunzip_device_file(sys.argv[1].encode(), b'/dev/null', 0, 0, 0, st)\n"
        "    assert rc == 0, L.nh_last_error()\n"
        "print('pieces %%d chunks %%d redecoded %%d host pieces %%d members %%d text %%d' %% tuple(st[:6]))\n") % ROOT
for f in (bg, gzp):
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, NOHUMAN_TRACE="1"), capture_output=True, text=True)
    print(os.path.basename(f), os.path.getsize(f), r.stdout.strip())
    print("\n".join(l[l.index(":", 40) + 2:] for l in r.stderr.splitlines() if "gzip reader on GPU" in l)[-700:], flush=True)
for f in (plain, bg, gzp):
    os.remove(f)
