"""PCIe-inclusive rate of the host-buffer entry nh_classify_batch ("PCIe note"):
H2D of the bases + k_classify + D2H of the records, blocking, per call.  bench.py's `value` is the
device-resident entry; this is the number a host that owns its I/O sees.
    python tools/host_batch_bench.py [pairs=1000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nohuman_amd import Engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 150
cap = 1_431_655_765
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
rng = np.random.default_rng(1)
n_seq = 2 * pairs
offs = (np.arange(n_seq + 1, dtype=np.uint64) * L)
pageable = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_seq * L)].copy()
pinned_t = torch.empty(n_seq * L, dtype=torch.uint8).pin_memory()
pinned = pinned_t.numpy()
pinned[:] = pageable
for label, buf in (("pageable host buffer", pageable), ("pinned host buffer", pinned)):
    for rep in range(4):
        t = time.time()
        res = eng.classify(buf, offs, True, 0.0)
        dt = time.time() - t
    print("%-22s %7.2f ms per %d pairs  = %7.1f Mreads/s  (%.1f GB/s of bases)" % (label, dt * 1e3, pairs, 2 * pairs / dt / 1e6, n_seq * L / dt / 1e9))
eng.close()
