"""The HYBRID gzip reader (round 6): with the kept text re-encoded on the GPU the chip's codec kernels are what
nh_run waits for while the host's cores idle, so some cells of each input's piece grid can be inflated by host workers
(RangeGunzip, tests/test_gunzip_ranges.py) beside the GPU's -- decoded ahead, taken in stream order, their text uploaded and
indexed on the device like any other piece's.  It is correct and it does not pay (profiles/r06_hybrid.txt: the chip ends up
waiting for the host's cells), so it is an option (NOHUMAN_GZ_HYBRID=n), not the default.  Whatever the mix, the run writes what the host reader writes (the reference hands
its inputs to the path as they are, /root/reference/src/main.rs:267), the members' CRCs are checked across pieces of both kinds,
and damage fails the run."""
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "toy_db")


def _fastq(seed, n, tag=b""):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        ln = int(rng.choice([150, 150, 150, 75, 251, 36]))
        s = bytes(acgt[rng.integers(0, 4, ln)])
        q = bytes((rng.integers(0, 40, ln) + 33).astype(np.uint8))
        out.append(b"@read.%d%s some words\n%s\n+\n%s\n" % (i, tag, s, q))
    return b"".join(out)


def _run(tmp_path, name, in1, in2=None, **kw):
    from nohuman_amd import Engine
    o1, o2, k = tmp_path / (name + "_o1"), tmp_path / (name + "_o2"), tmp_path / (name + "_k")
    with Engine.open(DB) as eng:
        st = eng.run(str(in1), str(o1), in2=str(in2) if in2 else None, out2=str(o2) if in2 else None, kraken_output=str(k), threads=8, **kw)
    rd = (lambda p: gzip.decompress(p.read_bytes())) if kw.get("out_codec") == 2 else (lambda p: p.read_bytes())
    return (rd(o1), rd(o2) if in2 else b"", k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))


def _scale(monkeypatch, seg=65536, stretch=4096, batch=512, hybrid="3"):
    monkeypatch.setenv("NOHUMAN_GZDEV_MIN_BYTES", "0")
    monkeypatch.setenv("NOHUMAN_GZDEV_ROOM", str(8 << 20))
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", str(seg))
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", str(stretch))
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", str(batch))
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    monkeypatch.setenv("NOHUMAN_GZ_HYBRID", hybrid)


def _host_pieces(err):
    import re
    return [int(x) for x in re.findall(r"(\d+) pieces inflated by the host's cores", err)]


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("codec", [0, 2])
@pytest.mark.parametrize("seg,stretch", [(65536, 4096), (16384, 4096), (262144, 8192)])
def test_hybrid_reader_writes_what_the_host_reader_writes(tmp_path, monkeypatch, capfd, paired, codec, seg, stretch):
    d1, d2 = _fastq(1, 9000, b"/1"), _fastq(2, 9000, b"/2")
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(d1, 6))
    p2.write_bytes(gzip.compress(d2, 1))
    _scale(monkeypatch, seg, stretch)
    monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
    hyb = _run(tmp_path, "hyb", p1, p2 if paired else None, out_codec=codec)
    err = capfd.readouterr().err
    got = _host_pieces(err)
    assert len(got) == (2 if paired else 1) and min(got) >= 1, err[-3000:]  # the host's cores really took pieces of every input
    assert "hybrid on" in err
    monkeypatch.setenv("NOHUMAN_GZ_HYBRID", "0")
    dev = _run(tmp_path, "dev", p1, p2 if paired else None, out_codec=codec)
    assert not _host_pieces(capfd.readouterr().err)
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p1, p2 if paired else None, out_codec=codec)
    assert hyb == host == dev
    assert host[3][0] == 9000 and host[0] == d1 and (not paired or host[1] == d2)  # (random reads: every one is kept)


def test_members_flush_points_and_stored_blocks_across_pieces_of_both_kinds(tmp_path, monkeypatch, capfd):
    """gzip members that end inside cells and at their edges, pigz-style flush points (empty stored blocks), a stretch of
    incompressible qualities (stored blocks: nothing for either search to find), an empty member at the end"""
    rng = np.random.default_rng(5)
    data = _fastq(3, 6000)
    cuts = sorted(int(x) for x in rng.integers(0, len(data), 7)) + [len(data)]
    parts, pos = [], 0
    for i, c in enumerate(cuts):
        blk = data[pos:c]
        if i % 3 == 1:  # flush points every few KB, as pigz / gzp write them
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            out = []
            for j in range(0, len(blk), 7000):
                out.append(co.compress(blk[j:j + 7000]))
                out.append(co.flush(zlib.Z_SYNC_FLUSH))
            out.append(co.flush())
            parts.append(b"".join(out))
        else:
            parts.append(gzip.compress(blk, int(rng.choice([1, 6, 9]))))
        pos = c
    noisy = b"".join(b"@n.%d\n%s\n+\n%s\n" % (i, b"ACGT" * 40, bytes((rng.integers(0, 60, 160) + 33).astype(np.uint8))) for i in range(2000))
    parts.append(gzip.compress(noisy, 1))
    parts.append(gzip.compress(b""))
    p = tmp_path / "m.fq.gz"
    p.write_bytes(b"".join(parts))
    want = data + noisy
    for seg, stretch in ((32768, 4096), (131072, 4096)):
        _scale(monkeypatch, seg, stretch, batch=300)
        monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
        hyb = _run(tmp_path, "hyb", p)
        assert min(_host_pieces(capfd.readouterr().err)) >= 1
        monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
        host = _run(tmp_path, "host", p)
        capfd.readouterr()
        assert hyb == host and host[0] == want


def test_damage_fails_the_run_with_the_hybrid_reader(tmp_path, monkeypatch):
    from nohuman_amd import EngineError
    good = gzip.compress(_fastq(4, 8000), 6)
    _scale(monkeypatch)
    monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
    for where in (len(good) // 4, len(good) // 2, len(good) * 3 // 4, len(good) - 6):
        bad = bytearray(good)
        bad[where] ^= 0x5A
        p = tmp_path / "bad.fq.gz"
        p.write_bytes(bytes(bad))
        with pytest.raises(EngineError):
            _run(tmp_path, "bad", p)
    p = tmp_path / "cut.fq.gz"
    p.write_bytes(good[: len(good) * 2 // 3])
    with pytest.raises(EngineError):
        _run(tmp_path, "cut", p)


def test_a_wrong_crc_in_a_piece_of_either_kind_fails_the_run(tmp_path, monkeypatch):
    """the members' books are kept across pieces of both kinds: a CRC that comes out wrong from the 6th piece on (the file is
    fine) is seen at the member's end whoever decoded that piece, and ends the run once records are out"""
    from nohuman_amd import EngineError
    p = tmp_path / "r.fq.gz"
    p.write_bytes(gzip.compress(_fastq(6, 9000), 6))
    _scale(monkeypatch, 16384, 4096, batch=100)
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_CRC", "6")
    with pytest.raises(EngineError) as ei:
        _run(tmp_path, "x", p, out_codec=2)
    assert "crc error" in str(ei.value)


def test_the_hybrid_is_off_unless_asked_for(tmp_path, monkeypatch, capfd):
    """measured and left off (profiles/r06_hybrid.txt): a run without NOHUMAN_GZ_HYBRID has no host lane, whatever its outputs"""
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(_fastq(7, 3000, b"/1"), 6))
    p2.write_bytes(gzip.compress(_fastq(8, 3000, b"/2"), 6))
    _scale(monkeypatch)
    monkeypatch.delenv("NOHUMAN_GZ_HYBRID")
    monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
    for codec in (2, 0):
        _run(tmp_path, "x", p1, p2, out_codec=codec)
        err = capfd.readouterr().err
        assert "hybrid on" not in err and not _host_pieces(err)
    monkeypatch.setenv("NOHUMAN_GZ_HYBRID", "1")  # "1": the run's threads less four, shared between the files
    _run(tmp_path, "y", p1, p2, out_codec=2)
    assert "hybrid on (2 host workers per file)" in capfd.readouterr().err
