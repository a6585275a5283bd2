"""Soak of the gzip reader on the GPU against zlib: random FASTQ-like texts (several kinds of ids, read lengths, quality
alphabets, low-complexity stretches, binary and constant runs), random zlib levels / strategies / flush points / member cuts,
random piece and chunk sizes of the reader.  Every case: bytes == the text that was compressed; a mismatch stops the run
and leaves the input behind.   python tools/gunzip_soak.py [cases=200] [seed=1] [max_mb=24]"""
import ctypes as C, os, sys, tempfile, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nohuman_amd import _lib
L = _lib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_mb = int(sys.argv[3]) if len(sys.argv) > 3 else 24


def text_of(rng, n):
    out, size, i = [], 0, int(rng.integers(0, 10**6))
    kind = int(rng.integers(0, 5))
    rl = int(rng.choice([36, 75, 100, 150, 151, 250, 1000, 9000]))
    qa = int(rng.choice([2, 4, 8, 12, 40]))
    inst = b"A0%d:%d:H%dXX" % (rng.integers(100, 999), rng.integers(1, 400), rng.integers(10**4, 10**5))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    while size < n:
        r = rng.random()
        if r < 0.002:
            rec = bytes(rng.integers(0, 256, int(rng.integers(1, 70000)), dtype=np.uint8))  # binary: stored blocks
        elif r < 0.004:
            rec = bytes([int(rng.integers(32, 127))]) * int(rng.integers(1, 300000))  # a run: far beyond 16 : 1
        else:
            ln = rl if kind != 4 else int(rng.integers(20, 2 * rl))
            if rng.random() < 0.03:
                seq = bytes(acgt[rng.integers(0, 4, 3)]) * (ln // 3 + 1)
                seq = seq[:ln]
            else:
                seq = bytes(acgt[rng.integers(0, 4, ln)])
            q0 = rng.integers(0, qa, ln)
            if kind in (1, 3):
                q0 = np.sort(q0)[::-1]
            qual = bytes((q0 + 35).astype(np.uint8))
            if kind == 0:
                hdr = b"@read.%d some description" % i
            elif kind == 1:
                hdr = b"@%s:%d:%d:%d 1:N:0:ACGTAC" % (inst, i // 10000 % 8 + 1, 1000 + i // 100 % 2000, 1000 + i * 7 % 30000)
            elif kind == 2:
                hdr = b"@SRR%d.%d %d length=%d" % (seed + 1000000, i, i, ln)
            else:
                hdr = b"@%x-%x" % (int(rng.integers(0, 2**40)), i)
            rec = hdr + b"\n" + seq + (b"\r\n+\r\n" if kind == 3 and rng.random() < 0.5 else b"\n+\n") + qual + b"\n"
        out.append(rec)
        size += len(rec)
        i += 1
    return b"".join(out)[:n]


def bgzf_of(rng, data):
    import struct
    out = []
    block = int(rng.choice([65280, 65280, 20000, 3000, 500]))
    for i in list(range(0, len(data), block)) + ([None] if rng.random() < 0.7 or not data else []):
        blk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(int(rng.choice([1, 6, 9])), zlib.DEFLATED, -15)
        body = co.compress(blk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) +
                   body + struct.pack("<II", zlib.crc32(blk), len(blk)))
    return b"".join(out)


def gz_of(rng, data):
    if rng.random() < 0.15 and len(data) < 12_000_000:
        raw = bgzf_of(rng, data)
        if rng.random() < 0.2:
            raw += gzip_tail(rng)
        return raw
    parts, pos = [], 0
    nmem = int(rng.choice([1, 1, 1, 2, 3, 9]))
    cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, nmem - 1)) + [len(data)]
    for cut in cuts:
        blk = data[pos:cut]
        pos = cut
        level = int(rng.choice([1, 2, 4, 6, 6, 6, 9]))
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 6 + [zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED]))
        co = zlib.compressobj(level, zlib.DEFLATED, 31, int(rng.choice([8, 9])), strat)
        if rng.random() < 0.15 and len(blk):
            step = int(rng.integers(5000, 400000))
            fl = int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            buf = []
            for o in range(0, len(blk), step):
                buf.append(co.compress(blk[o:o + step]))
                buf.append(co.flush(fl))
            parts.append(b"".join(buf) + co.flush())
        else:
            parts.append(co.compress(blk) + co.flush())
    return b"".join(parts)


TAIL = b""


def gzip_tail(rng):
    return b""  # (a BGZF file followed by other members changes the text: the caller would have to know; kept simple)


rng = np.random.default_rng(seed)
tmp = tempfile.mkdtemp(prefix="nh_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
src, dst = os.path.join(tmp, "x.gz"), os.path.join(tmp, "x.out")
st = (C.c_uint64 * 8)()
t0 = time.time()
tot = host = redo = 0
for k in range(cases):
    n = int(rng.choice([0, 1, 100, 70000, 10**6, 5 * 10**6, int(rng.integers(1, max_mb * 10**6))]))
    data = text_of(rng, n) if n else b""
    raw = gz_of(rng, data)
    if rng.random() < 0.1:
        raw += bytes(int(rng.integers(1, 5000)))  # trailing zeros are ignored like gzip does
    seg = int(rng.choice([0, 0, 1 << 20, 300000, 64000, 8 << 20]))
    stretch = int(rng.choice([0, 0, 32768, 16384, 8192, 4096, 2048]))
    open(src, "wb").write(raw)
    rc = L.nh_gunzip_device_file(src.encode(), dst.encode(), 0, seg, stretch, st)
    got = open(dst, "rb").read() if rc == 0 else None
    if rc != 0 or got != data:
        keep = os.path.join(ROOT, "gpurun_out", "soak_fail_%d_%d.gz" % (seed, k))
        os.makedirs(os.path.dirname(keep), exist_ok=True)
        open(keep, "wb").write(raw)
        print("CASE %d FAILED: rc %d %s; text %d bytes, gzip %d, seg %d stretch %d -> %s" % (
            k, rc, L.nh_last_error() if rc else ("%d bytes back" % len(got)), len(data), len(raw), seg, stretch, keep), flush=True)
        sys.exit(1)
    tot += len(data)
    host += st[3]
    redo += st[2]
    if k % 20 == 19:
        print("%d cases, %.2f GB of text, %d pieces by the host decoder, %d chunks decoded again, %.0f s" % (k + 1, tot / 1e9, host, redo, time.time() - t0), flush=True)
print("OK: %d cases, %.2f GB of text, %d pieces by the host decoder, %d chunks decoded again" % (cases, tot / 1e9, host, redo))
