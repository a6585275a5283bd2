"""Differential soak of nh_run's two gzip readers: random FASTQ files with every oddity of the record semantics (CRLF, "+id"
lines, trailing blanks on any line, lower case, N, empty sequences, a missing final newline, a truncated last record, an empty
header line in the middle -- kraken2 stops there --, mates of unequal record counts), gzip at random levels and member cuts,
random batch / piece / chunk sizes, one to three lanes of the reader (pieces decoded ahead), the hybrid reader's host lane on and off, single-end and paired, plain and gzip outputs, classified-out and unclassified-out:
the reader on the GPU (inflate + record index there) must write exactly what the host reader writes.
    python tools/run_soak.py [cases=200] [seed=1]"""
import gzip, os, sys, tempfile, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nohuman_amd import Engine
from tests import synth
DB = os.path.join(ROOT, "tests", "golden", "toy_db")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
import json
genomes = None


def fastq(rng, n, odd, long_tail=False):
    out = []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    n_long = int(rng.integers(3, 12)) if long_tail and n else 0  # short reads first, then very long ones
    for i in range(n + n_long):
        ln = int(rng.choice([0, 20, 35, 60, 100, 150, 151, 300, 2000])) if rng.random() < 0.3 else 150
        if i >= n:
            ln = int(rng.integers(20000, 120000))
        seq = bytes(acgt[rng.integers(0, 4, ln)])
        if odd and rng.random() < 0.1:
            seq = seq.lower()
        if odd and ln and rng.random() < 0.1:
            k = int(rng.integers(0, ln))
            seq = seq[:k] + b"N" * int(rng.integers(1, 5)) + seq[k + 1:]
        qual = bytes((rng.integers(0, 40, len(seq)) + 33).astype(np.uint8))
        hdr = b"@r%d" % i + (b" extra words %d" % rng.integers(0, 99) if rng.random() < 0.5 else b"") + (b"/1" if rng.random() < 0.2 else b"")
        eol = b"\r\n" if odd and rng.random() < 0.15 else b"\n"
        pad = lambda: (b" " * int(rng.integers(1, 3)) if odd and rng.random() < 0.05 else b"") + (b"\t" if odd and rng.random() < 0.02 else b"")
        plus = b"+" + (hdr[1:] if odd and rng.random() < 0.2 else b"")
        out.append(hdr + pad() + eol + seq + pad() + eol + plus + eol + qual + pad() + eol)
    data = b"".join(out)
    if odd and n and rng.random() < 0.15:
        data = data[:-1] if data.endswith(b"\n") else data  # no final newline
    if odd and n > 3 and rng.random() < 0.1:
        data = data[: len(data) - int(rng.integers(1, 200))]  # the last record is cut short
    if odd and n > 10 and rng.random() < 0.08:
        k = int(rng.integers(1, n - 1))
        cut = sum(len(x) for x in out[:k])
        data = data[:cut] + (b"\n" if rng.random() < 0.5 else b"@\n") + data[cut:]  # an empty header line: the input ends there
    elif odd and n > 10 and rng.random() < 0.04:
        k = int(rng.integers(1, n - 1))
        cut = sum(len(x) for x in out[:k])
        data = data[:cut] + bytes(rng.choice(list(b"X+>#"), 1).astype(np.uint8)) + b"\n" + data[cut:]  # a one-character line that is no '@': malformed on any reader
    return data


def bgzf(rng, data):
    import struct
    out = []
    block = int(rng.choice([65280, 20000, 3000]))
    for i in list(range(0, len(data), block)) + [None]:
        blk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(int(rng.choice([1, 6])), zlib.DEFLATED, -15)
        body = co.compress(blk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) +
                   body + struct.pack("<II", zlib.crc32(blk), len(blk)))
    return b"".join(out)


def gz(rng, data):
    if rng.random() < 0.2:
        return bgzf(rng, data)  # bgzip's format: the reader on the GPU takes its chunk starts from the members' headers
    parts, pos = [], 0
    cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, int(rng.choice([0, 0, 1, 3])))) + [len(data)]
    for c in cuts:
        parts.append(gzip.compress(data[pos:c], int(rng.choice([1, 6, 9]))))
        pos = c
    return b"".join(parts)


tmp = tempfile.mkdtemp(prefix="nh_runsoak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
eng = Engine.open(DB)
bad = 0
for k in range(cases):
    paired = bool(rng.random() < 0.5)
    odd = bool(rng.random() < 0.7)
    n = int(rng.choice([0, 1, 7, 300, 2000, 9000]))
    long_tail = bool(rng.random() < 0.25)
    d1 = fastq(rng, n, odd, long_tail)
    d2 = fastq(rng, n if rng.random() < 0.8 else max(0, n - int(rng.integers(0, 5))), odd, long_tail) if paired else None
    f1, f2 = os.path.join(tmp, "a_1.fq.gz"), os.path.join(tmp, "a_2.fq.gz")
    open(f1, "wb").write(gz(rng, d1))
    if paired:
        open(f2, "wb").write(gz(rng, d2))
    kw = dict(keep_human=bool(rng.random() < 0.3), out_codec=int(rng.choice([0, 0, 2])), confidence=float(rng.choice([0.0, 0.2])), threads=4)
    want_k = bool(rng.random() < 0.5)
    os.environ["NOHUMAN_BATCH_FRAGS"] = str(int(rng.choice([16, 64, 500, 4096])))
    os.environ["NOHUMAN_GZDEV_SEG"] = str(int(rng.choice([16384, 65536, 1 << 20])))
    os.environ["NOHUMAN_GZDEV_STRETCH"] = str(int(rng.choice([1024, 2048, 8192])))
    lanes = int(rng.choice([1, 1, 2, 3]))  # several lanes of the reader on the one device: pieces decoded ahead of the stream
    os.environ.pop("NOHUMAN_GZ_LANES", None)
    if lanes > 1:
        os.environ["NOHUMAN_GZ_LANES"] = str(lanes)
        os.environ["NOHUMAN_GZDEV_STRETCH"] = str(int(rng.choice([4096, 8192])))
        os.environ["NOHUMAN_GZDEV_SEG"] = str(int(rng.choice([16384, 65536, 262144])))
    # the hybrid reader (round 6): some cells of the stream inflated by host workers beside the GPU's (NH_SOAK_HYBRID=n forces it)
    hyb = int(os.environ.get("NH_SOAK_HYBRID", "-1"))
    if hyb < 0:
        hyb = int(rng.choice([0, 0, 2, 3]))
    os.environ["NOHUMAN_GZ_HYBRID"] = str(hyb)
    # the product's default (no reader named: it may hand the file over to the host reader mid-stream, and single-end batches
    # are cut by text) in the cases with the long tail and in a third of the others; the reader named in the rest
    auto = long_tail or bool(rng.random() < 0.33)
    os.environ["NOHUMAN_GZDEV_MIN_BYTES"] = "0"
    os.environ.pop("NOHUMAN_GZDEV_ROOM", None)
    os.environ.pop("NOHUMAN_BATCH_TEXT", None)
    if long_tail:
        os.environ["NOHUMAN_GZDEV_ROOM"] = str(int(rng.choice([1 << 20, 2 << 20, 8 << 20])))
        os.environ["NOHUMAN_BATCH_TEXT"] = str(int(rng.choice([100000, 1 << 20, 512 << 20])))
    res = {}
    for reader in ("device", "host"):
        os.environ["NOHUMAN_GZ_READER"] = reader
        if auto and reader == "device":
            os.environ.pop("NOHUMAN_GZ_READER")
        o1, o2, ko = (os.path.join(tmp, "%s_%s" % (reader, x)) for x in ("o1", "o2", "k"))
        for o in (o1, o2, ko):
            if os.path.exists(o):
                os.remove(o)
        try:
            st = eng.run(f1, o1, in2=f2 if paired else None, out2=o2 if paired else None, kraken_output=ko if want_k else None, **kw)
            rd = (lambda p: gzip.decompress(open(p, "rb").read())) if kw["out_codec"] == 2 else (lambda p: open(p, "rb").read())
            res[reader] = (rd(o1), rd(o2) if paired else b"", open(ko, "rb").read() if want_k else b"", st.total_sequences, st.classified, st.total_bases)
        except Exception as ex:
            res[reader] = ("ERROR", str(ex).split(":")[-1].strip()[:60] if "malformed" not in str(ex) else "malformed")
    if res["device"] != res["host"]:
        bad += 1
        keep = os.path.join(ROOT, "gpurun_out", "runsoak_fail_%d_%d" % (seed, k))
        os.makedirs(keep, exist_ok=True)
        open(os.path.join(keep, "a_1.fq.gz"), "wb").write(open(f1, "rb").read())
        if paired:
            open(os.path.join(keep, "a_2.fq.gz"), "wb").write(open(f2, "rb").read())
        dv, hv = res["device"], res["host"]
        print("CASE %d DIFFERS: paired %s odd %s n %d long tail %s default reader %s kw %s batch %s seg %s stretch %s lanes %d room %s hybrid %d" % (k, paired, odd, n, long_tail, auto, kw,
              os.environ["NOHUMAN_BATCH_FRAGS"], os.environ["NOHUMAN_GZDEV_SEG"], os.environ["NOHUMAN_GZDEV_STRETCH"], lanes, os.environ.get("NOHUMAN_GZDEV_ROOM"), hyb))
        print("   device:", [x if not isinstance(x, bytes) else (len(x), zlib.crc32(x)) for x in dv])
        print("   host  :", [x if not isinstance(x, bytes) else (len(x), zlib.crc32(x)) for x in hv], flush=True)
        if bad >= 5:
            break
    if k % 25 == 24:
        print("%d cases, %d differ" % (k + 1, bad), flush=True)
eng.close()
print("DONE: %d cases, %d differ" % (cases, bad))
sys.exit(1 if bad else 0)
