"""Debugging aid: decodes a gzip file with the device reader under several settings (NOHUMAN_GZDEV_NOCRC: the text is
written even when a member's CRC fails) and reports where the bytes differ from the host decoder's.
    python tools/gz_debug.py file.gz"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = sys.argv[1]
ref = src + ".host"
code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
        "from nohuman_amd import _lib\n"
        "L = _lib.lib(); st = (C.c_uint64 * 8)()\n"
        "if sys.argv[3] == 'host':\n"
        "    rc = L.nh_gunzip_file(sys.argv[1].encode(), sys.argv[2].encode(), 16, 0, st)\n"
        "else:\n"
        "    rc = L.nh_gunzip_device_file(sys.argv[1].encode(), sys.argv[2].encode(), 0, int(sys.argv[3]), int(sys.argv[4]), st)\n"
        "print('rc', rc, L.nh_last_error() if rc else '', list(st))\n") % ROOT
subprocess.run([sys.executable, "-c", code, src, ref, "host"], check=True)
a = np.memmap(ref, np.uint8, "r")
print("host text", a.size, flush=True)
runs = [({}, 0, 0), ({}, 0, 16384), ({}, 64 << 20, 32768)]
for env, seg, stretch in runs:
    out = src + ".dev"
    e = dict(os.environ, NOHUMAN_GZDEV_NOCRC="1", NOHUMAN_TRACE="1", **env)
    r = subprocess.run([sys.executable, "-c", code, src, out, str(seg), str(stretch)], env=e, capture_output=True, text=True)
    print("==", env, seg, stretch, r.stdout.strip())
    if os.path.isdir("gpurun_out"):
        open("gpurun_out/gz_debug_pieces_%d.log" % runs.index((env, seg, stretch)), "w").write(r.stderr)
    print("\n".join(l[:400] for l in r.stderr.splitlines() if "gzdev] member" in l or "gzip reader" in l or "rror" in l))
    b = np.memmap(out, np.uint8, "r")
    print("sizes", a.size, b.size)
    n = min(a.size, b.size)
    nbad, spans = 0, []
    for o in range(0, n, 1 << 28):
        hi = min(n, o + (1 << 28))
        d = np.nonzero(a[o:hi] != b[o:hi])[0]
        if d.size and not nbad and a.size != b.size:  # text is missing: where does the device's text go on in the host's?
            x = o + int(d[0])
            key = bytes(b[x:x + 400])
            hay = bytes(a[x:x + (1 << 28)])
            print("first difference at", x, "; the device's next 400 bytes are found in the host's text at +", hay.find(key))
            nbad = -1
            break
        if d.size:
            nbad += d.size
            if len(spans) < 12:
                cuts = np.nonzero(np.diff(d) > 64)[0]
                starts = np.concatenate(([d[0]], d[cuts + 1]))
                ends = np.concatenate((d[cuts], [d[-1]]))
                for s_, e_ in list(zip(starts, ends))[:12 - len(spans)]:
                    spans.append((int(o + s_), int(o + e_)))
    print("differing bytes", nbad, "first spans", spans, flush=True)
    for s_, e_ in spans[:3]:
        print("  host:", bytes(a[max(0, s_ - 40):s_ + 80]))
        print("  dev :", bytes(b[max(0, s_ - 40):s_ + 80]))
    del b
    os.remove(out)
