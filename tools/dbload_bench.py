"""Time of nh_open on an HPRC.r2-sized database directory (reported separately from the run, as
kraken2's own timer excludes the database load): 5.7 GB hash.k2d from the page cache into HBM.
    TMPDIR=/dev/shm python tools/dbload_bench.py"""
import os, sys, time, tempfile, shutil, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nohuman_amd import Engine

cap = 1_431_655_765
tmp = tempfile.mkdtemp(prefix="nh_db_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
    info = eng.info
    open(os.path.join(tmp, "opts.k2d"), "wb").write(eng.opts_image())
    open(os.path.join(tmp, "taxo.k2d"), "wb").write(eng.taxonomy_image())
    with open(os.path.join(tmp, "hash.k2d"), "wb") as f:
        f.write(struct.pack("<4Q", info.capacity, info.size, info.key_bits, info.value_bits))
        eng.download_table().tofile(f)
    eng.close()
    size = os.path.getsize(os.path.join(tmp, "hash.k2d"))
    for rep in range(3):
        t = time.time()
        e = Engine.open(tmp)
        dt = time.time() - t
        e.close()
        print("nh_open: %.2f s for hash.k2d of %.2f GB = %.2f GB/s" % (dt, size / 1e9, size / dt / 1e9))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
