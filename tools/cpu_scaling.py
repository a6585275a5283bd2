"""How does the CPU oracle scale with threads on this box? (informs bench.py's cpu_baseline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from oracle import minidb
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; free -g | head -2")
# synthetic table on the host: 1.43e9 cells (5.7 GB), iid fill at load 0.7 (not a real build; same probe cost class)
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1_431_655_765
rng = np.random.default_rng(1)
t0 = time.time()
cells = rng.integers(1, 1 << 32, size=cap, dtype=np.uint32)
mask = rng.random(cap, dtype=np.float32) < 0.3
cells[mask] = 0
cells |= 31  # value bits non-zero where occupied
cells[mask] = 0
print("table built in %.1fs" % (time.time() - t0))
import struct
from tests import synth
tax = minidb.Taxonomy({i + 1: i for i in range(30)})
odb = orc.OracleDB(minidb.opts_bytes(), tax.to_bytes(), cells=cells, header=(cap, int((~mask).sum()), 27, 5))
n = 400000
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * 300)]
offs = np.arange(2 * n + 1, dtype=np.uint64) * 150
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    m = n if th > 4 else n // 8
    t0 = time.perf_counter()
    out, lk = odb.classify(bases[:m * 300], offs[:2 * m + 1], True, 0.0, threads=th)
    dt = time.perf_counter() - t0
    print("threads %3d: %.3f Mreads/s (%.2fs, %d reads, %.1f lookups/read)" % (th, 2 * m / dt / 1e6, dt, 2 * m, lk.sum() / (2 * m)))
