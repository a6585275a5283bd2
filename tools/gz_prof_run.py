"""One pass of the GPU gzip reader over a prepared .gz file (for rocprofv3): python3 tools/gz_prof_run.py file.gz [seg] [stretch] [room]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nohuman_amd import _lib  # noqa: E402
L = _lib.lib()
st = (C.c_uint64 * 8)()
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
stretch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 4:
    os.environ.setdefault("NOHUMAN_GZDEV_ROOM", sys.argv[4])
rc = L.nh_gunzip_device_file(sys.argv[1].encode(), b"/dev/null", 0, seg, stretch, st)
print("rc", rc, "pieces", st[0], "chunks", st[1], "text", st[5])
