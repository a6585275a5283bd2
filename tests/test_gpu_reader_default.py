"""The DEFAULT gzip input path of nh_run (no NOHUMAN_GZ_READER) must finish whatever the host reader finishes, with the same
bytes (the reference hands inputs verbatim to the path, /root/reference/src/main.rs:267, where kraken2
reads any valid FASTQ): short reads followed by very long ones (length-sorted long-read files, adapter dimers first) --
single-end batches are cut by text and a piece hands out every complete record, paired batches that outgrow the room the
reader keeps in front of a piece go to the host reader from that record on, with ONE warning line and no error."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "toy_db")


def _short_then_long(seed, n_short=2000, n_long=24, lo=50_000, hi=300_000, tag=b""):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n_short):
        s = bytes(acgt[rng.integers(0, 4, 150)])
        out.append(b"@short.%d%s\n%s\n+\n%s\n" % (i, tag, s, b"I" * 150))
    for i in range(n_long):
        ln = int(rng.integers(lo, hi))
        s = bytes(acgt[rng.integers(0, 4, ln)])
        out.append(b"@long.%d%s\n%s\n+\n%s\n" % (i, tag, s, b"5" * ln))
    for i in range(50):  # and short ones again behind them
        s = bytes(acgt[rng.integers(0, 4, 150)])
        out.append(b"@tail.%d%s\n%s\n+\n%s\n" % (i, tag, s, b"I" * 150))
    return b"".join(out)


def _run(tmp_path, name, in1, in2=None, **kw):
    from nohuman_amd import Engine
    o1, o2, k = tmp_path / (name + "_o1"), tmp_path / (name + "_o2"), tmp_path / (name + "_k")
    with Engine.open(DB) as eng:
        st = eng.run(str(in1), str(o1), in2=str(in2) if in2 else None, out2=str(o2) if in2 else None, kraken_output=str(k), threads=4, **kw)
    return (o1.read_bytes(), o2.read_bytes() if in2 else b"", k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))


def _small_scale(monkeypatch, room=4 << 20, batch=4096):
    # the product's sizes (pieces of 512 MiB, 768 MiB in front of a piece, batches of 262144 records) scaled down
    monkeypatch.setenv("NOHUMAN_GZDEV_MIN_BYTES", "0")
    monkeypatch.setenv("NOHUMAN_GZDEV_ROOM", str(room))
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "65536")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "4096")
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", str(batch))
    monkeypatch.setenv("NOHUMAN_TRACE", "1")


def test_single_end_short_reads_then_long_reads_stay_on_the_gpu(tmp_path, monkeypatch, capfd):
    data = _short_then_long(1)
    p = tmp_path / "ont.fq.gz"
    p.write_bytes(gzip.compress(data, 6))
    _small_scale(monkeypatch)
    monkeypatch.setenv("NOHUMAN_BATCH_TEXT", str(1 << 20))  # (the product's 512 MiB, scaled: batches are cut by text)
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    dev = _run(tmp_path, "dev", p)
    err = capfd.readouterr().err
    assert "gzip reader on GPU" in err and "WARN" not in err, err[-2000:]  # the reader on the GPU did all of it
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p)
    assert dev == host
    assert host[3][0] == 2000 + 24 + 50 and host[0] == data  # (random reads: nothing is classified, everything is kept)


@pytest.mark.parametrize("batch", [64, 4096])
def test_paired_batches_that_outgrow_the_room_go_to_the_host_reader(tmp_path, monkeypatch, capfd, batch):
    d1, d2 = _short_then_long(2, tag=b"/1"), _short_then_long(3, tag=b"/2")
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(d1, 6))
    p2.write_bytes(gzip.compress(d2, 1))
    _small_scale(monkeypatch, batch=batch)
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    dev = _run(tmp_path, "dev", p1, p2)
    err = capfd.readouterr().err
    assert err.count("the host reader goes on from record") >= 1, err[-2000:]
    assert "holds more text than" in err
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p1, p2)
    assert dev == host and host[3][0] == 2074 and host[0] == d1 and host[1] == d2
    # asked for by name there is no silent change of reader: the run fails and says why
    from nohuman_amd import EngineError
    monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
    with pytest.raises(EngineError) as ei:
        _run(tmp_path, "named", p1, p2)
    assert "holds more text than" in str(ei.value)


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("at", [1, 2, 5])
def test_the_reader_on_the_gpu_hands_over_mid_stream(tmp_path, monkeypatch, capfd, paired, at):
    """Whatever stops the device reader after it has handed out batches (here: a test knob before its `at`-th piece) the run
    goes on with the host reader behind the records already handed out: same bytes, same counts."""
    raw1, raw2 = _short_then_long(7, 6000, 0, tag=b"/1"), _short_then_long(8, 6000, 0, tag=b"/2")  # (some forty pieces a file)
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(raw1, 6))
    p2.write_bytes(gzip.compress(raw2, 6))
    _small_scale(monkeypatch, batch=100)
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "16384")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "2048")
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    monkeypatch.setenv("NOHUMAN_GZDEV_FAIL_AT", str(at))
    dev = _run(tmp_path, "dev", p1, p2 if paired else None)
    err = capfd.readouterr().err
    assert "the host reader goes on from record" in err, err[-2000:]
    monkeypatch.delenv("NOHUMAN_GZDEV_FAIL_AT")
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p1, p2 if paired else None)
    assert dev == host and host[3][0] == raw1.count(b"\n") // 4


@pytest.mark.parametrize("paired", [False, True])
def test_a_wrong_decode_seen_only_by_the_crc_fails_the_run_once_records_are_out(tmp_path, monkeypatch, capfd, paired):
    """the member's CRC-32 is the only end-to-end check on the device inflate, and a one-member file of
    many pieces has its early pieces handed out, classified and written before the trailer is read.  A wrong decode that only
    the CRC sees (NOHUMAN_GZDEV_FAKE_CRC=k: from the k-th piece on the text's CRC comes out wrong; the file itself is fine,
    so the host reader would pass ITS check) must fail the run -- round 5 handed the file to the host reader and returned
    NH_OK with the suspect records already in the output.  With nothing handed out yet (the whole file in one piece: the
    trailer is read before the first batch goes out) the handover stays: the host reader starts from the top."""
    from nohuman_amd import EngineError
    raw1, raw2 = _short_then_long(7, 6000, 0, tag=b"/1"), _short_then_long(8, 6000, 0, tag=b"/2")  # (some forty pieces a file)
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(raw1, 6))
    p2.write_bytes(gzip.compress(raw2, 6))
    _small_scale(monkeypatch, batch=100)
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "16384")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "2048")
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_CRC", "9")
    with pytest.raises(EngineError) as ei:
        _run(tmp_path, "dev", p1, p2 if paired else None)
    assert "crc error" in str(ei.value), str(ei.value)
    assert "the host reader goes on from record" not in capfd.readouterr().err
    # one piece holds the whole file: the check fails before a record has gone out, and the host reader reads the (good) file
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", str(64 << 20))
    monkeypatch.setenv("NOHUMAN_GZDEV_ROOM", str(64 << 20))
    monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_CRC", "1")
    dev = _run(tmp_path, "dev1", p1, p2 if paired else None)
    assert "the host reader goes on from record 0" in capfd.readouterr().err
    monkeypatch.delenv("NOHUMAN_GZDEV_FAKE_CRC")
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p1, p2 if paired else None)
    assert dev == host and host[3][0] == raw1.count(b"\n") // 4


def test_a_one_character_header_line_is_malformed_on_both_readers(tmp_path, monkeypatch):
    """the host parser (and kraken2) end the input at an empty line or a lone '@', anything else without '@' is
    malformed -- a one-character line too; the record kernel used to end the input there silently."""
    from nohuman_amd import EngineError
    raw = open(os.path.join(ROOT, "tests", "golden", "reads_se.fq"), "rb").read()
    cut = raw.index(b"\n@", len(raw) // 2) + 1
    p = tmp_path / "x.fq.gz"
    _small_scale(monkeypatch)
    for junk, ends in ((b"X\nACGT\n+\nIIII\n", False), (b"@\nACGT\n+\nIIII\n", True), (b"\n", True), (b" \t\n", True), (b"X \n", False)):
        p.write_bytes(gzip.compress(raw[:cut] + junk + raw[cut:]))
        res = {}
        for reader in ("device", "host"):
            monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
            try:
                res[reader] = _run(tmp_path, reader, p)
            except EngineError as ex:
                res[reader] = "malformed" if "malformed FASTQ file (exp. '@', saw \"X\")" in str(ex) else str(ex)
        assert res["device"] == res["host"], junk
        assert (res["host"] != "malformed") == ends, junk
        if ends:
            assert res["host"][3][0] == raw[:cut].count(b"\n") // 4


def test_buffers_kept_between_runs_are_bounded_by_bytes(tmp_path):
    """the store of idle reader buffers is bounded by bytes and evicts oldest first; runs over inputs of
    other sizes replace what is kept instead of piling it up.  (A child process: the bound is read once.)"""
    import subprocess
    import sys
    raw = open(os.path.join(ROOT, "tests", "golden", "reads_se.fq"), "rb").read()
    for k in range(4):
        (tmp_path / ("x%d.fq.gz" % k)).write_bytes(gzip.compress(raw * (k + 1)))
    code = """
import os, sys
sys.path.insert(0, %r)
from nohuman_amd import Engine, _lib
L = _lib.lib()
seen = []
with Engine.open(%r) as eng:
    for k, room in enumerate([64 << 20, 160 << 20, 400 << 20, 96 << 20]):
        os.environ["NOHUMAN_GZDEV_ROOM"] = str(room)  # buffers of another size every run
        eng.run(os.path.join(%r, "x%%d.fq.gz" %% k), os.path.join(%r, "o%%d.fq" %% k), threads=4)
        seen.append(int(L.nh_cache_bytes(0, 0)))
    print("SEEN", *seen)
print("AFTER", int(L.nh_cache_bytes(0, 0)), int(L.nh_cache_bytes(0, 1)))
""" % (ROOT, DB, str(tmp_path), str(tmp_path))
    env = dict(os.environ, NOHUMAN_GZDEV_MIN_BYTES="0", NOHUMAN_GZDEV_CACHE_GB="0.5")
    env.pop("NOHUMAN_GZ_READER", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    seen = [int(x) for x in out.stdout.split("SEEN")[1].split("\n")[0].split()]
    after = [int(x) for x in out.stdout.split("AFTER")[1].split()]
    assert all(0 < x <= (512 << 20) for x in seen), seen  # kept, and never beyond the bound (two text buffers of 400 MiB do not fit: the newest stays)
    assert after == [0, 0]  # nh_close empties the store


def test_fasta_in_gzip_goes_to_the_host_parser_in_the_default_mode_too(tmp_path, monkeypatch, capfd):
    """No reader named, the GPU reader chosen by size: a gzip file that holds FASTA (wrapped lines) is handed to the host parser
    before a batch has gone out -- no warning, same bytes as the host reader."""
    from tests.fastq_util import read_fastq
    rs = read_fastq(os.path.join(ROOT, "tests", "golden", "reads_se.fq"))
    fa = b"".join(b">" + r[1] + b" d\n" + b"".join(r[2][j:j + 60] + b"\n" for j in range(0, len(r[2]), 60)) for r in rs) * 3
    p = tmp_path / "x.fa.gz"
    p.write_bytes(gzip.compress(fa, 6))
    _small_scale(monkeypatch, batch=100)
    monkeypatch.delenv("NOHUMAN_GZ_READER", raising=False)
    dev = _run(tmp_path, "dev", p)
    err = capfd.readouterr().err
    assert "gzip reader: GPU" in err and "WARN" not in err, err[-1500:]
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    host = _run(tmp_path, "host", p)
    assert dev == host and host[3][0] == 3 * len(rs)


@pytest.mark.parametrize("shape", ["both host", "plain + gzip on the GPU", "gzip on the GPU + plain"])
@pytest.mark.parametrize("codec", [0, 2])
def test_paired_halves_of_unequal_length_are_used_in_parts(tmp_path, monkeypatch, shape, codec):
    """A reader that has to cut a batch by TEXT (32-bit text offsets; here a 30 KB budget), or two files on different readers,
    hand the pairing halves of different record counts.  That used to stop the run ("paired inputs lost step ... lower
    NOHUMAN_BATCH_FRAGS"); the shorter half decides now and the rest of the longer one pairs with the other file's next half.
    Mates of different read lengths, so that the two files' text cuts never coincide."""
    raw1 = open(os.path.join(ROOT, "tests", "golden", "reads_pe_1.fq"), "rb").read() * 5
    recs2 = []
    from tests.fastq_util import read_fastq
    for k in range(5):
        for i, (h, _id, seq, q) in enumerate(read_fastq(os.path.join(ROOT, "tests", "golden", "reads_pe_2.fq"))):
            cut = 60 + (i * 7) % 90  # mate 2: 60-149 bases
            recs2.append(h + b"\n" + seq[:cut] + b"\n+\n" + q[:cut] + b"\n")
    raw2 = b"".join(recs2)
    gz1, gz2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    pl1, pl2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
    gz1.write_bytes(gzip.compress(raw1, 6)); gz2.write_bytes(gzip.compress(raw2, 6))
    pl1.write_bytes(raw1); pl2.write_bytes(raw2)
    _small_scale(monkeypatch, batch=64)
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "16384")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "2048")
    monkeypatch.setenv("NOHUMAN_GZ_READER", "host")
    want = _run(tmp_path, "want", gz1, gz2, out_codec=0)  # whole halves, host reader, plain outputs: the reference bytes
    monkeypatch.setenv("NOHUMAN_BATCH_TEXT", "30000")
    if shape == "both host":
        a, b = gz1, gz2
    else:
        monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
        a, b = (pl1, gz2) if shape.startswith("plain") else (gz1, pl2)
    from nohuman_amd import Engine
    o1, o2, k = tmp_path / "o1", tmp_path / "o2", tmp_path / "k"
    with Engine.open(DB) as eng:
        st = eng.run(str(a), str(o1), in2=str(b), out2=str(o2), kraken_output=str(k), threads=4, out_codec=codec, codec_threads=2)
    rd = (lambda p: gzip.decompress(p.read_bytes())) if codec == 2 else (lambda p: p.read_bytes())
    got = (rd(o1), rd(o2), k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))
    assert got == want
    assert want[3][0] == raw1.count(b"\n") // 4
