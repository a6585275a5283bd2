"""How long does opening a full-size database directory take (files in the page cache / tmpfs)?  Builds the synthetic
HPRC.r2-sized table, writes it out as a database directory, then times Engine.open on it.
usage: db_load_bench.py [capacity=1431655765]"""
import os, struct, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nohuman_amd import Engine
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1_431_655_765
tmp = tempfile.mkdtemp(prefix="nh_db_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
    info = eng.info
    db = os.path.join(tmp, "db")
    os.makedirs(db)
    open(os.path.join(db, "opts.k2d"), "wb").write(eng.opts_image())
    open(os.path.join(db, "taxo.k2d"), "wb").write(eng.taxonomy_image())
    with open(os.path.join(db, "hash.k2d"), "wb") as f:
        f.write(struct.pack("<4Q", info.capacity, info.size, info.key_bits, info.value_bits))
        eng.download_table().tofile(f)
    eng.close()
    size = os.path.getsize(os.path.join(db, "hash.k2d"))
    for rep in range(int(os.environ.get("DB_LOAD_REPS", "3"))):
        t = time.perf_counter()
        e = Engine.open(db)
        dt = time.perf_counter() - t
        print("Engine.open: %.3f s for a %.2f GB hash.k2d = %.2f GB/s (copies in HBM: %s)" % (dt, size / 1e9, size / dt / 1e9, os.environ.get("NOHUMAN_TABLE_COPIES", "default 4")), flush=True)
        e.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
