"""End-to-end timing of nh_run (files in, files out) on synthetic FASTQ -- SURVEY.md section 8d
"two timings, never mixed": this is the e2e one (reader + H2D + classify + D2H + writer), next to
bench.py's kernel-resident number.  Builds a synthetic database directory on local disk first.
    python tools/e2e_bench.py [pairs=2000000] [capacity=134217728]"""
import gzip, os, sys, time, tempfile, shutil, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nohuman_amd import Engine

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 134_217_689
tmp = tempfile.mkdtemp(prefix="nh_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    t0 = time.time()
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
    rng = np.random.default_rng(3)
    L = 150
    def write_fastq(path, tag):
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(pairs, L))]
        with open(path, "wb") as f:
            block = 100000
            for b0 in range(0, pairs, block):
                n = min(block, pairs - b0)
                rows = []
                for i in range(n):
                    rows.append(b"@syn.%d/%d\n" % (b0 + i, tag))
                    rows.append(seq[b0 + i].tobytes())
                    rows.append(b"\n+\n" + b"I" * L + b"\n")
                f.write(b"".join(rows))
    f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
    write_fastq(f1, 1); write_fastq(f2, 2)
    print("setup %.1fs: db %.2f GB, 2 x %.0f MB FASTQ" % (time.time() - t0, cap * 4 / 1e9, os.path.getsize(f1) / 1e6))
    with Engine.open(db) as e:
        for label, a, b in (("plain PE", f1, f2),):
            for rep in range(2):
                t = time.time()
                st = e.run(a, os.path.join(tmp, "o_1.fq"), in2=b, out2=os.path.join(tmp, "o_2.fq"))
                dt = time.time() - t
                print("%s run %d: %.2fs wall, %.2f Mreads/s e2e (%d fragments, %d classified)" % (label, rep, dt, 2 * st.total_sequences / dt / 1e6, st.total_sequences, st.classified))
        t = time.time()
        st = e.run(f1, os.path.join(tmp, "o.fq"))
        dt = time.time() - t
        print("plain SE: %.2fs wall, %.2f Mreads/s e2e" % (dt, st.total_sequences / dt / 1e6))
        t = time.time()
        os.system("gzip -1 -k %s %s" % (f1, f2))
        print("gzip -1 took %.1fs" % (time.time() - t))
        t = time.time()
        st = e.run(f1 + ".gz", os.path.join(tmp, "o_1.fq"), in2=f2 + ".gz", out2=os.path.join(tmp, "o_2.fq"))
        dt = time.time() - t
        print("gzip PE: %.2fs wall, %.2f Mreads/s e2e" % (dt, 2 * st.total_sequences / dt / 1e6))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
