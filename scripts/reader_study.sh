#!/bin/bash
# input side alone on the GPU box's host cores: inflate only, inflate + parse, plain read + parse
REPO=$(pwd); D=/dev/shm/nh_rs; mkdir -p $D
g++ -O3 -std=c++17 -Inohuman_amd/csrc tools/reader_bench.cpp nohuman_amd/csrc/nh_inflate.cpp nohuman_amd/csrc/nh_fastx.cpp -lz -lpthread -ldl -o /tmp/nh_reader_bench || exit 1
python3 - <<PY
import numpy as np, os, sys
sys.path.insert(0, "$REPO")
from nohuman_amd import _lib
n=4_000_000; L=150
rng=np.random.default_rng(1)
hdr=b"@syn.000000000/1\n"
reclen=len(hdr)+L+3+L+1
rec=np.empty((n,reclen),dtype=np.uint8)
rec[:,:len(hdr)]=np.frombuffer(hdr,dtype=np.uint8)
idx=np.arange(n)
for d in range(9):
    rec[:,5+8-d]=48+(idx//10**d)%10
p=len(hdr)
rec[:,p:p+L]=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,(n,L))]
rec[:,p+L:p+L+3]=np.frombuffer(b"\n+\n",dtype=np.uint8)
rec[:,p+L+3:p+2*L+3]=73
rec[:,p+2*L+3]=10
rec.tofile("$D/r.fq")
assert _lib.lib().nh_compress_file(b"$D/r.fq", b"$D/r.fq.gz", 2, 16)==0
import time, ctypes as C
for th in (1,4,8,16):
    st=(C.c_uint64*3)()
    t=time.time(); _lib.lib().nh_gunzip_file(b"$D/r.fq.gz", b"/dev/null", th, 0, st); dt=time.time()-t
    print("inflate only, %2d threads: %.3f s = %.2f GB/s of text"%(th, dt, n*reclen/dt/1e9))
PY
for th in 0; do /tmp/nh_reader_bench $D/r.fq $th | tail -1; done
for th in 1 4 8 16; do /tmp/nh_reader_bench $D/r.fq.gz $th | tail -1; done
rm -rf $D
