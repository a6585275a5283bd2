"""Which gzip reader should a run START on?  nh_run end to end on gzip FASTQ of growing size (paired 150 bp, and the ONT shape
in one file), the reader on the GPU against the host reader, outputs gzip / plain / none -- the numbers behind
device_reader_pays() (nh_run.hip) and profiles/r05_reader_choice.txt.
    python tools/reader_choice.py [reps=3] [threads=16]"""
import os, sys, time, tempfile, shutil, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nohuman_amd import Engine, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tmp = tempfile.mkdtemp(prefix="nh_choice_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
L = _lib.lib()
try:
    cap = 134_217_689
    eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=7)
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base_pairs = 1_000_000

    def write_pe(path, tag):
        seq = acgt[rng.integers(0, 4, size=(base_pairs, 150))]
        with open(path, "wb") as f:
            for b0 in range(0, base_pairs, 100000):
                rows = []
                for i in range(b0, min(base_pairs, b0 + 100000)):
                    rows.append(b"@syn.%d/%d\n" % (i, tag))
                    rows.append(seq[i].tobytes())
                    rows.append(b"\n+\n" + b"I" * 150 + b"\n")
                f.write(b"".join(rows))

    f1, f2 = os.path.join(tmp, "b_1.fq"), os.path.join(tmp, "b_2.fq")
    write_pe(f1, 1)
    write_pe(f2, 2)
    for f in (f1, f2):
        assert L.nh_compress_file(os.fsencode(f), os.fsencode(f + ".gz"), 2, threads) == 0
        os.remove(f)
    g1, g2 = open(f1 + ".gz", "rb").read(), open(f2 + ".gz", "rb").read()
    print("base: %d pairs, %.1f MB of gzip a file" % (base_pairs, len(g1) / 1e6), flush=True)

    def ont(n):
        lens = np.clip(np.exp(rng.normal(8.8, 0.85, n)), 200, 200000).astype(np.int64)
        total = int(lens.sum())
        seq = acgt[rng.integers(0, 4, size=total, dtype=np.uint8)]
        fo = os.path.join(tmp, "ont.fq")
        with open(fo, "wb") as f:
            off = 0
            for i in range(n):
                ln = int(lens[i])
                f.write(b"@ont.%d\n" % i)
                f.write(seq[off:off + ln].tobytes())
                f.write(b"\n+\n" + b"5" * ln + b"\n")
                off += ln
        assert L.nh_compress_file(os.fsencode(fo), os.fsencode(fo + ".gz"), 2, threads) == 0
        os.remove(fo)
        return fo + ".gz"

    def bench(label, a, b, kw):
        o1, o2 = os.path.join(tmp, "o1"), os.path.join(tmp, "o2")
        res = {"device": [], "host": []}
        n = 0
        for rep in range(reps + 1):
            for reader in ("device", "host"):
                os.environ["NOHUMAN_GZ_READER"] = reader
                for o in (o1, o2):
                    if os.path.exists(o):
                        os.remove(o)
                t = time.perf_counter()
                st = eng.run(a, o1, in2=b, out2=o2 if b else None, threads=threads, **kw)
                dt = time.perf_counter() - t
                n = st.total_sequences
                if rep:
                    res[reader].append(dt)
        os.environ.pop("NOHUMAN_GZ_READER")
        d, h = sorted(res["device"])[len(res["device"]) // 2], sorted(res["host"])[len(res["host"]) // 2]
        print("%-44s device %.3f s   host %.3f s   device/host %.2f   (%d fragments)" % (label, d, h, d / h, n), flush=True)

    outs = (("gzip out", dict(out_codec=2, codec_threads=8)), ("plain out", dict()), ("nothing kept", dict(keep_human=True)))
    big = len(sys.argv) > 3 and sys.argv[3] == "big"  # the larger files, outputs written by the host only
    for k in ((30, 40, 60, 80) if big else (1, 2, 4, 8, 14, 20)):
        p1, p2 = os.path.join(tmp, "r_1.fq.gz"), os.path.join(tmp, "r_2.fq.gz")
        with open(p1, "wb") as fo:
            for _ in range(k):
                fo.write(g1)
        with open(p2, "wb") as fo:
            for _ in range(k):
                fo.write(g2)
        for what, kw in (outs[1:2] if big else outs):
            bench("PE %4.0f MB a file, %s" % (len(g1) * k / 1e6, what), p1, p2, kw)
        for what, kw in (outs[1:2] if big else outs[:1] + outs[2:]):
            bench("SE %4.0f MB, %s" % (len(g1) * k / 1e6, what), p1, None, kw)
    if big:
        raise SystemExit(0)
    for n in (100_000, 300_000):
        p = ont(n)
        for what, kw in outs:
            bench("ONT %d reads, %.0f MB, %s" % (n, os.path.getsize(p) / 1e6, what), p, None, kw)
    eng.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
