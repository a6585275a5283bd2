"""Host logic without a GPU: the FASTA/FASTQ reader of nh_run (nh_fastx_scan entry of the C ABI)
against a plain-Python statement of kraken2's record semantics."""
import bz2
import ctypes as C
import gzip
import os

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def fnv(parts):
    h = 0xcbf29ce484222325
    for p in parts:
        for b in p:
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        h = ((h ^ 0) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def py_records(data: bytes):
    """kraken2 BatchSequenceReader semantics, line by line (getline + StripString)."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # a final newline does not start another line
    i, recs, fmt = 0, [], None
    while i < len(lines):
        h = lines[i].rstrip()
        i += 1
        if fmt is None:
            fmt = "fq" if h[:1] == b"@" else "fa" if h[:1] == b">" else None
            if fmt is None:
                raise ValueError("unrecognized file format")
        if fmt == "fq":
            if not h:
                break
            if h[:1] != b"@":
                raise ValueError("malformed FASTQ")
            if len(h) <= 1 or i + 2 >= len(lines):  # sequence, '+' and quality lines must exist
                break
            seq, qual = lines[i].rstrip(), lines[i + 2].rstrip()
            i += 3
            recs.append((h, seq, qual))
        else:
            if h[:1] != b">":
                raise ValueError("malformed FASTA")
            if len(h) <= 1:
                break
            seq = b""
            while i < len(lines) and lines[i][:1] != b">":
                seq += lines[i].rstrip()
                i += 1
            recs.append((h, seq, b""))
    return recs


def scan(path):
    from nohuman_amd import _lib
    L = _lib.lib()
    n, nb, d = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.nh_fastx_scan(os.fsencode(path), C.byref(n), C.byref(nb), C.byref(d))
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())
    return n.value, nb.value, d.value


def expect(data):
    recs = py_records(data)
    return len(recs), sum(len(r[1]) for r in recs), fnv([x for r in recs for x in r])


CASES = {
    "plain": b"@r1 desc\nACGT\n+\nIIII\n@r2\nGGCC\n+r2\nJJJJ\n",
    "no_final_newline": b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+\nJJ",
    "crlf_and_trailing_ws": b"@r1 x \r\nACGT \r\n+\r\nIIII\t\r\n",
    "blank_line_ends_file": b"@r1\nACGT\n+\nIIII\n\n@r2\nGG\n+\nJJ\n",
    "truncated_last_record": b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+\n",
    "empty_sequence": b"@r1\n\n+\n\n@r2\nAC\n+\nII\n",
    "fasta_multiline": b">s1 first\nACGT\nTTGG\n\n>s2\nA\n>s3\n",
    "fasta_no_final_newline": b">s1\nACGT\nTT",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_reader_edge_cases(tmp_path, name):
    p = tmp_path / (name + ".txt")
    p.write_bytes(CASES[name])
    assert scan(p) == expect(CASES[name])


@pytest.mark.parametrize("chunk,batch", [(1, 1), (7, 2), (64, 3), (5, 4096)])
@pytest.mark.parametrize("name", sorted(CASES))
def test_reader_records_across_read_and_batch_boundaries(tmp_path, monkeypatch, name, chunk, batch):
    """The reader parses in place in the buffer it reads into: tiny reads and tiny batches put every
    kind of boundary (inside a line, between lines of a record, between records) at a buffer edge."""
    monkeypatch.setenv("NOHUMAN_READ_CHUNK", str(chunk))
    monkeypatch.setenv("NOHUMAN_SCAN_BATCH", str(batch))
    p = tmp_path / (name + ".txt")
    p.write_bytes(CASES[name])
    assert scan(p) == expect(CASES[name])
    gz = tmp_path / (name + ".gz")
    with gzip.open(gz, "wb") as f:
        f.write(CASES[name])
    assert scan(gz) == expect(CASES[name])


def test_reader_long_fasta_record_over_many_reads(tmp_path, monkeypatch):
    monkeypatch.setenv("NOHUMAN_READ_CHUNK", "1000")
    body = b"".join(b"ACGTTGCA" * 10 + b"\n" for _ in range(2000))  # one 160 kb record, 80-column lines
    data = b">chr1 test\n" + body + b">chr2\nAC\n"
    p = tmp_path / "long.fa"
    p.write_bytes(data)
    assert scan(p) == expect(data)
    assert expect(data)[:2] == (2, 160002)


def test_reader_golden_fixtures_plain_gzip_bzip2(tmp_path):
    data = open(os.path.join(GOLD, "reads_se.fq"), "rb").read() * 30  # 3.8 MB: several refills of the inflate buffer
    want = expect(data)
    plain = tmp_path / "r.fq"
    plain.write_bytes(data)
    gz = tmp_path / "r.fq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(data)
    bz = tmp_path / "r.fq.bz2"
    bz.write_bytes(bz2.compress(data))
    assert scan(plain) == want
    assert scan(gz) == want
    assert scan(bz) == want
    assert want[0] == 406 * 30


def test_reader_errors(tmp_path):
    p = tmp_path / "bad.txt"
    p.write_bytes(b"hello\nworld\n")
    with pytest.raises(RuntimeError) as ei:
        scan(p)
    assert "unrecognized file format" in str(ei.value)
    p.write_bytes(b"@r1\nACGT\n+\nIIII\nACGT\n")
    with pytest.raises(RuntimeError) as ei:
        scan(p)
    assert "malformed FASTQ" in str(ei.value)
    with pytest.raises(RuntimeError):
        scan(tmp_path / "missing.fq")


def test_bzip2_input_damage_is_an_error_not_a_short_read(tmp_path):
    """a truncated or corrupt .bz2 used to end the stream quietly (popen + ignored exit status):
    reads went missing from a decontamination run without a word.  The in-process decoder reports it;
    concatenated streams still decode like `bzip2 -dc`."""
    data = open(os.path.join(GOLD, "reads_se.fq"), "rb").read() * 10
    whole = bz2.compress(data)
    two = tmp_path / "two.fq.bz2"
    two.write_bytes(bz2.compress(data[:len(data) // 2], 1) + bz2.compress(data[len(data) // 2:], 9))
    assert scan(two) == expect(data)
    cut = tmp_path / "cut.fq.bz2"
    cut.write_bytes(whole[:len(whole) // 2])
    with pytest.raises(RuntimeError) as ei:
        scan(cut)
    assert "bzip2" in str(ei.value)
    # bytes behind the last stream that are no bzip2 stream: `bzip2 -dc` warns and still delivers the data
    tg = tmp_path / "garbage.fq.bz2"
    tg.write_bytes(whole + b"\x00" * 100 + b"not a stream")
    assert scan(tg) == expect(data)
    tg.write_bytes(whole + b"BZ")
    assert scan(tg) == expect(data)
    bad = bytearray(whole)
    bad[len(bad) // 2] ^= 0x55
    (tmp_path / "bad.fq.bz2").write_bytes(bytes(bad))
    with pytest.raises(RuntimeError) as ei:
        scan(tmp_path / "bad.fq.bz2")
    assert "bzip2" in str(ei.value)
