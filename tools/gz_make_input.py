"""Writes bench.py's FASTQ text as a gzip file (the library's block-parallel level-6 encoder): python3 tools/gz_make_input.py out.gz [records] [members] [mate tag 1|2]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nohuman_amd import _lib
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
members = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tag = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cx = types.SimpleNamespace(torch=torch, dev=torch.device("cuda", 0))
L = _lib.lib()
with open(out, "wb") as f:
    for k in range(members):
        plain = out + ".plain"
        bench.e2e_member(cx, n, 150, tag, k, plain)
        assert L.nh_compress_file(os.fsencode(plain), os.fsencode(plain + ".gz"), 2, 16) == 0
        f.write(open(plain + ".gz", "rb").read())
        os.remove(plain)
        os.remove(plain + ".gz")
print(out, os.path.getsize(out))
