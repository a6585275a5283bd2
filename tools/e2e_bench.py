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
    if os.environ.get("E2E_ONT"):  # configs[3] shape, scaled: long single-end reads, N50 ~ 10 kb
        from nohuman_amd import _lib
        n = int(os.environ["E2E_ONT"])
        threads = int(os.environ.get("E2E_THREADS", "16"))
        lens = np.clip(np.exp(rng.normal(8.8, 0.85, n)), 200, 200000).astype(np.int64)
        total = int(lens.sum())
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=total, dtype=np.uint8)]
        fo = os.path.join(tmp, "ont.fq")
        with open(fo, "wb") as f:
            off = 0
            for i in range(n):
                ln = int(lens[i])
                f.write(b"@ont.%d\n" % i)
                f.write(seq[off:off + ln].tobytes())
                f.write(b"\n+\n" + b"5" * ln + b"\n")
                off += ln
        del seq
        srt = np.sort(lens)[::-1]
        n50 = int(srt[np.searchsorted(np.cumsum(srt), total / 2)])
        print("setup %.1fs: %d reads, %.2f Gbases, N50 %d, FASTQ %.1f GB" % (time.time() - t0, n, total / 1e9, n50, os.path.getsize(fo) / 1e9))
        t = time.time()
        assert _lib.lib().nh_compress_file(os.fsencode(fo), os.fsencode(fo + ".gz"), 2, threads) == 0
        print("gzip took %.1fs, ratio %.2f" % (time.time() - t, os.path.getsize(fo) / os.path.getsize(fo + ".gz")))
        with Engine.open(db) as e:
            for label, path, codec in (("ONT plain", fo, 0), ("ONT plain", fo, 0), ("ONT gzip", fo + ".gz", 0),
                                       ("ONT gzip", fo + ".gz", 0), ("ONT gzip > gzip", fo + ".gz", 2), ("ONT gzip > gzip", fo + ".gz", 2)):
                out = os.path.join(tmp, "o.fq.gz" if codec else "o.fq")
                if os.path.exists(out):
                    os.remove(out)
                t = time.time()
                st = e.run(path, out, threads=threads, out_codec=codec, codec_threads=threads)
                dt = time.time() - t
                print("%-16s %6.2fs wall  %6.3f Mreads/s  %6.2f Gbases/s e2e  (%d reads, %d classified)" % (
                    label, dt, st.total_sequences / dt / 1e6, st.total_bases / dt / 1e9, st.total_sequences, st.classified), flush=True)
            # the gzip output (encoded on the GPU) inflates to the plain output
            import ctypes, filecmp
            back = os.path.join(tmp, "back.fq")
            st3 = (ctypes.c_uint64 * 3)()
            rc = _lib.lib().nh_gunzip_file(os.fsencode(os.path.join(tmp, "o.fq.gz")), os.fsencode(back), threads, 0, st3)
            print("gzip output: %.2f GB, ratio %.2f, inflates to the plain output: %s" % (
                os.path.getsize(os.path.join(tmp, "o.fq.gz")) / 1e9, os.path.getsize(back) / os.path.getsize(os.path.join(tmp, "o.fq.gz")),
                rc == 0 and filecmp.cmp(back, os.path.join(tmp, "o.fq"), shallow=False)))
        raise SystemExit(0)
    f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
    write_fastq(f1, 1); write_fastq(f2, 2)
    print("setup %.1fs: db %.2f GB, 2 x %.0f MB FASTQ" % (time.time() - t0, cap * 4 / 1e9, os.path.getsize(f1) / 1e6))
    reps = int(os.environ.get("E2E_REPS", "4"))  # the inputs are the generated files repeated `reps` times
    threads = int(os.environ.get("E2E_THREADS", str(min(16, os.cpu_count() or 1))))
    t = time.time()
    from nohuman_amd import _lib
    for f in (f1, f2):  # block-parallel gzip level 6 (nh_compress_file): one ordinary gzip member
        rc = _lib.lib().nh_compress_file(os.fsencode(f), os.fsencode(f + ".gz"), 2, threads)
        assert rc == 0
    print("gzip (nh_compress_file, %d threads) took %.1fs, ratio %.2f" % (threads, time.time() - t, os.path.getsize(f1) / os.path.getsize(f1 + ".gz")))
    def rep_file(path):
        if reps == 1:
            return path
        out = path + ".x%d" % reps
        with open(out, "wb") as fo:
            data = open(path, "rb").read()
            for _ in range(reps):
                fo.write(data)
        return out
    gz_only = bool(os.environ.get("E2E_GZ_ONLY"))
    G1, G2 = rep_file(f1 + ".gz"), rep_file(f2 + ".gz")
    F1, F2 = (f1, f2) if gz_only else (rep_file(f1), rep_file(f2))
    o1, o2 = os.path.join(tmp, "o_1.fq"), os.path.join(tmp, "o_2.fq")
    with Engine.open(db) as e:
        def go(label, a, b, th):
            for p in (o1, o2):
                if os.path.exists(p):
                    os.remove(p)
            t = time.time()
            st = e.run(a, o1, in2=b, out2=o2 if b else None, threads=th)
            dt = time.time() - t
            n = st.total_sequences * (2 if b else 1)
            print("%-28s %6.2fs wall  %7.2f Mreads/s e2e  (%d reads, %d classified)" % (label, dt, n / dt / 1e6, n, st.classified))
        if gz_only:  # full-scale configs[2] run: gzip pairs only, outputs kept in tmpfs
            go("gzip PE, %d threads" % threads, G1, G2, threads)
            go("gzip PE, %d threads" % threads, G1, G2, threads)
            raise SystemExit(0)
        go("plain PE (warm-up)", F1, F2, threads)
        go("plain PE", F1, F2, threads)
        go("plain SE", F1, None, threads)
        go("gzip PE, 1 thread (zlib-like)", G1, G2, 1)
        go("gzip PE, %d threads" % threads, G1, G2, threads)
        go("gzip SE, %d threads" % threads, G1, None, threads)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
