"""What a COLD `nohuman` process costs (the CLI host, one run per process: what a user does) against the warm second run of a
process that bench.py's e2e leg times: DB load, buffer set-up (page-locking, HBM), the run itself.  NOHUMAN_TRACE stage lines kept.
    python tools/cli_cold.py [pairs_per_member=5000000] [members=4]"""
import os, shutil, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nohuman_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
members = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = tempfile.mkdtemp(prefix="nh_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    for t in (1, 2):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gz_make_input.py"), os.path.join(tmp, "r_%d.fq.gz" % t), str(n), str(members)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cap = 1_431_655_765
    eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
    db = os.path.join(tmp, "db")
    os.makedirs(db)
    open(os.path.join(db, "opts.k2d"), "wb").write(eng.opts_image())
    open(os.path.join(db, "taxo.k2d"), "wb").write(eng.taxonomy_image())
    i = eng.info
    with open(os.path.join(db, "hash.k2d"), "wb") as f:
        f.write(struct.pack("<4Q", i.capacity, i.size, i.key_bits, i.value_bits))
        eng.download_table().tofile(f)
    eng.close()
    del eng
    exe = os.path.join(ROOT, "nohuman_amd", "bin", "nohuman")
    for what, outs in (("gzip out", ("o_1.fq.gz", "o_2.fq.gz")), ("plain out", ("o_1.fq", "o_2.fq"))):
        for rep in range(2):
            for o in outs:
                p = os.path.join(tmp, o)
                if os.path.exists(p):
                    os.remove(p)
            t = time.perf_counter()
            r = subprocess.run([exe, "-t", "16", "--db", db, "-o", os.path.join(tmp, outs[0]), "-O", os.path.join(tmp, outs[1]),
                                os.path.join(tmp, "r_1.fq.gz"), os.path.join(tmp, "r_2.fq.gz")], env=dict(os.environ, NOHUMAN_TRACE="1"),
                               capture_output=True, text=True, cwd=tmp)
            dt = time.perf_counter() - t
            print("== %s, process %d: rc %d, wall %.3f s = %.1f Mreads/s" % (what, rep, r.returncode, dt, 2 * n * members / dt / 1e6))
            for ln in r.stderr.splitlines():
                if os.environ.get("CLI_COLD_ALL") and rep == 0 and "piece " not in ln or any(k in ln for k in ("wall", "nh_run:", "ERROR", "WARN")):
                    print("   ", ln[:230])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
