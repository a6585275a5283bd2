"""The gzip reader on the GPU (nh_gunzip.hip) on bench.py's own FASTQ text: stage times of the device reader alone
(nh_gunzip_device_file, NOHUMAN_TRACE kernel milliseconds per stage) and, with `e2e`, nh_run on a gzip pair with the
device reader and with the host reader (NOHUMAN_GZ_READER=host), outputs compared.
    python tools/gunzip_dev_bench.py [records=3000000] [members=2] [e2e]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nohuman_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 2
e2e = "e2e" in sys.argv
L = _lib.lib()
cx = types.SimpleNamespace(torch=torch, dev=torch.device("cuda", 0))
tmp = tempfile.mkdtemp(prefix="nh_gzdev_", dir="/dev/shm")
try:
    files = {}
    t0 = time.time()
    for tag in (1, 2) if e2e else (1,):
        gz = os.path.join(tmp, "r_%d.fq.gz" % tag)
        text = 0
        with open(gz, "wb") as out:
            for k in range(members):
                plain = os.path.join(tmp, "m.fq")
                text += bench.e2e_member(cx, n, 150, tag, k, plain)
                assert L.nh_compress_file(os.fsencode(plain), os.fsencode(plain + ".gz"), 2, 16) == 0
                out.write(open(plain + ".gz", "rb").read())
                os.remove(plain)
                os.remove(plain + ".gz")
        files[tag] = (gz, text)
    gz, text = files[1]
    print("setup %.1f s: %d members of %d records, %.2f GB of text, %.3f GB of gzip (%.2f : 1)" % (
        time.time() - t0, members, n, text / 1e9, os.path.getsize(gz) / 1e9, text / os.path.getsize(gz)), flush=True)
    # ---- the device reader alone, through the C ABI in a child (NOHUMAN_TRACE prints the kernel times at close)
    code = ("import sys, time, ctypes as C; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from nohuman_amd import _lib\n"
            "L = _lib.lib(); st = (C.c_uint64 * 8)()\n"
            "for rep in range(2):\n"
            "    t = time.perf_counter()\n"
            "    rc = L.nh_gunzip_device_file(sys.argv[1].encode(), b'/dev/null', 0, int(sys.argv[2]), int(sys.argv[3]), st)\n"
            "    dt = time.perf_counter() - t\n"
            "    assert rc == 0, L.nh_last_error()\n"
            "print('seg %%s stretch %%s: wall %%.3f s = %%.2f GB/s of text; pieces %%d chunks %%d redecoded %%d host pieces %%d' %% (\n"
            "      sys.argv[2], sys.argv[3], dt, st[5] / dt / 1e9, st[0], st[1], st[2], st[3]))\n") % ROOT
    runs = ((0, 0, 0, 6 << 30), (256 << 20, 32768, 0, 3 << 30), (0, 0, 0, 256 << 20), (64 << 20, 32768, 0, 1 << 30), (256 << 20, 16384, 0, 3 << 30))
    if os.environ.get("GZDEV_BENCH_QUICK"):
        runs = ((256 << 20, 32768, 0, 3 << 30), (32 << 20, 32768, 0, 3 << 30))
    if os.environ.get("GZDEV_BENCH_QUICK") == "2":
        runs = ((0, 0, 0, 3 << 30),)
    if os.environ.get("GZDEV_BENCH_QUICK") == "3":  # piece and chunk sizes
        runs = ((256 << 20, 32768, 0, 6 << 30), (160 << 20, 32768, 0, 6 << 30), (320 << 20, 32768, 0, 6 << 30), (512 << 20, 32768, 0, 6 << 30),
                (512 << 20, 65536, 0, 6 << 30), (640 << 20, 65536, 0, 6 << 30), (384 << 20, 49152, 0, 6 << 30))
    for seg, stretch, v1, room in runs:
        env = dict(os.environ, NOHUMAN_TRACE="1", NOHUMAN_GZDEV_ROOM=str(room))
        print("room %d MiB" % (room >> 20))
        r = subprocess.run([sys.executable, "-c", code, gz, str(seg), str(stretch)], env=env, capture_output=True, text=True)
        print(r.stdout.strip())
        print("\n".join(l for l in r.stderr.splitlines() if "gzip reader" in l or "rror" in l or "gz prof" in l)[-2500:], flush=True)
    if e2e:
        from nohuman_amd import Engine
        import struct
        cap = 134_217_689
        eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
        outs = {}
        for mode in ("device", "host"):
            for codec, cname in ((0, "plain"), (2, "gzip")):
                o1, o2 = os.path.join(tmp, "o1_%s" % mode), os.path.join(tmp, "o2_%s" % mode)
                os.environ["NOHUMAN_GZ_READER"] = mode
                os.environ["NOHUMAN_TRACE"] = "1"
                t = time.perf_counter()
                st = eng.run(files[1][0], o1, in2=files[2][0], out2=o2, threads=16, out_codec=codec, codec_threads=8)
                dt = time.perf_counter() - t
                print("nh_run reader=%s out=%s: %.3f s = %.1f Mreads/s (%d pairs, %d classified)" % (
                    mode, cname, dt, 2 * st.total_sequences / dt / 1e6, st.total_sequences, st.classified), flush=True)
                if codec == 0:
                    h = subprocess.run(["md5sum", o1, o2], capture_output=True, text=True).stdout.split()
                    outs.setdefault(mode, (h[0], h[2]))
        print("outputs equal (device reader vs host reader):", outs.get("device") == outs.get("host"))
        eng.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
