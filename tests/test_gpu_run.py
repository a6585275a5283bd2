"""GPU end-to-end tests of the whole-run entry (nh_run through the C ABI and through the
CommandRunner mirror): FASTQ fixtures in, kraken2-compatible files out, checked against the golden
expectations (made by the closed-form Python restatement) and the record semantics of SURVEY.md A.6."""
import gzip
import json
import os

import pytest

from tests.fastq_util import read_fastq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DB = os.path.join(GOLD, "toy_db")


def _expected(name, conf):
    # the goldens exist for both ambiguity rules; a child run of this file under the other rule
    # (test_gpu_parity.py::test_parity_suites_under_the_other_ambiguity_rule) picks its own
    if os.environ.get("NOHUMAN_OPT_AMBIGUITY_RULE") == "0":
        name = name.replace(".json", "_rule0.json")
    exp = json.load(open(os.path.join(GOLD, name)))
    ext = exp["meta"]["external_ids"]
    recs = exp["records"]
    calls = [r["by_conf"][str(conf)][0] for r in recs]
    return exp, ext, recs, calls


def _fastq_bytes(recs, suffixes=None):
    out = []
    for i, (h, _rid, seq, quals) in enumerate(recs):
        sfx = suffixes[i] if suffixes else b""
        out.append(h + sfx + b"\n" + seq + b"\n+\n" + quals + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("keep_human", [False, True])
@pytest.mark.parametrize("conf", [0.0, 0.5])
def test_single_end_run(tmp_path, keep_human, conf):
    from nohuman_amd import Engine
    _, ext, recs, calls = _expected("expected_se.json", conf)
    inp = os.path.join(GOLD, "reads_se.fq")
    reads = read_fastq(inp)
    out = tmp_path / "kraken_out.fq"
    kout = tmp_path / "k.txt"
    with Engine.open(DB) as eng:
        st = eng.run(inp, str(out), kraken_output=str(kout), confidence=conf, keep_human=keep_human)
    n_class = sum(1 for c in calls if c)
    assert (st.total_sequences, st.classified, st.unclassified) == (len(reads), n_class, len(reads) - n_class)
    assert st.total_bases == sum(len(r[2]) for r in reads)
    keep = [i for i, c in enumerate(calls) if bool(c) == keep_human]
    sfx = [b" kraken:taxid|%d" % ext[calls[i]] if calls[i] else b"" for i in keep]
    assert out.read_bytes() == _fastq_bytes([reads[i] for i in keep], sfx)
    # kraken output: C/U, id, external taxid, length, hit list
    lines = kout.read_text().split("\n")[:-1]
    assert len(lines) == len(reads)
    for i, line in enumerate(lines):
        cu, rid, taxid, ln, hl = line.split("\t")
        assert cu == ("C" if calls[i] else "U")
        assert rid.encode() == reads[i][1]
        assert int(taxid) == (ext[calls[i]] if calls[i] else 0)
        assert int(ln) == len(reads[i][2])
        assert hl == recs[i]["hitlist"]


def test_paired_end_run_gzip_inputs(tmp_path):
    """paired inputs, one of them gzip-compressed (kraken2 reads .gz transparently), '#' naming."""
    from nohuman_amd import CommandRunner
    conf = 0.1
    _, ext, recs, calls = _expected("expected_pe.json", conf)
    in1 = os.path.join(GOLD, "reads_pe_1.fq")
    in2gz = tmp_path / "reads_pe_2.fq.gz"
    with open(os.path.join(GOLD, "reads_pe_2.fq"), "rb") as f, gzip.open(in2gz, "wb") as g:
        g.write(f.read())
    r1 = read_fastq(in1)
    r2 = read_fastq(os.path.join(GOLD, "reads_pe_2.fq"))
    kout = tmp_path / "k.txt"
    runner = CommandRunner("kraken2")
    assert runner.is_executable()
    # exactly the argv nohuman builds (/root/reference/src/main.rs:215-267)
    runner.run(["--threads", "2", "--db", DB, "--output", str(kout), "--confidence", str(conf),
                "--paired", "--unclassified-out", str(tmp_path / "kraken_out#.fq"), in1, str(in2gz)])
    st = runner.last_stats
    n_class = sum(1 for c in calls if c)
    assert (st.total_sequences, st.classified) == (len(r1), n_class)
    keep = [i for i, c in enumerate(calls) if not c]
    assert (tmp_path / "kraken_out_1.fq").read_bytes() == _fastq_bytes([r1[i] for i in keep])
    assert (tmp_path / "kraken_out_2.fq").read_bytes() == _fastq_bytes([r2[i] for i in keep])
    lines = kout.read_text().split("\n")[:-1]
    assert len(lines) == len(r1)
    for i, line in enumerate(lines):
        cu, rid, taxid, ln, hl = line.split("\t")
        assert cu == ("C" if calls[i] else "U")
        want_id = r1[i][1]
        if len(want_id) > 2 and want_id[-2:] in (b"/1", b"/2"):
            want_id = want_id[:-2]
        assert rid.encode() == want_id
        assert int(taxid) == (ext[calls[i]] if calls[i] else 0)
        assert ln == "%d|%d" % (len(r1[i][2]), len(r2[i][2]))
        assert hl == recs[i]["hitlist"]


@pytest.mark.parametrize("reader", ["device", "host"])
def test_gzip_pairs_through_the_reader_on_the_gpu_and_on_the_host(tmp_path, monkeypatch, capfd, reader):
    """gzip inputs through nh_run with the device reader on (NOHUMAN_GZ_READER=device: no silent
    change of reader) and off (=host): the same kept FASTQ bytes, kraken lines and counts, and they are the goldens'.
    Both mates gzip, several members in one of them, small pieces so that the chunks' windows are chained by the scan."""
    from nohuman_amd import Engine
    conf = 0.0
    _, ext, recs, calls = _expected("expected_pe.json", conf)
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read()
    raw2 = open(os.path.join(GOLD, "reads_pe_2.fq"), "rb").read()
    cut = raw1.index(b"\n@", len(raw1) // 2) + 1
    in1, in2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    in1.write_bytes(gzip.compress(raw1[:cut], 6) + gzip.compress(raw1[cut:], 9))
    in2.write_bytes(gzip.compress(raw2, 1))
    monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "16384")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "2048")
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    r1 = read_fastq(os.path.join(GOLD, "reads_pe_1.fq"))
    r2 = read_fastq(os.path.join(GOLD, "reads_pe_2.fq"))
    o1, o2, k = tmp_path / "o_1.fq", tmp_path / "o_2.fq", tmp_path / "k.txt"
    with Engine.open(DB) as eng:
        st = eng.run(str(in1), str(o1), in2=str(in2), out2=str(o2), kraken_output=str(k), confidence=conf, threads=4)
    err = capfd.readouterr().err
    assert ("gzip reader on GPU" in err) == (reader == "device"), err[-2000:]  # the reader that was asked for did the work
    keep = [i for i, c in enumerate(calls) if not c]
    assert (st.total_sequences, st.classified) == (len(r1), len(r1) - len(keep))
    assert o1.read_bytes() == _fastq_bytes([r1[i] for i in keep])
    assert o2.read_bytes() == _fastq_bytes([r2[i] for i in keep])
    lines = k.read_text().split("\n")[:-1]
    assert [ln.split("\t")[4] for ln in lines] == [r["hitlist"] for r in recs]


def _run_both_readers(tmp_path, monkeypatch, name, data1, data2=None, out_codec=0, batch=None, **kw):
    """nh_run on gzip inputs with the reader on the GPU (record index there, text resident in HBM), with the GPU inflating for
    the host parser ("device-text") and with the host reader: the three must write the same bytes and count the same"""
    from nohuman_amd import Engine
    outs = {}
    p1 = tmp_path / (name + "_1.fq.gz")
    p1.write_bytes(gzip.compress(data1, 6))
    p2 = None
    if data2 is not None:
        p2 = tmp_path / (name + "_2.fq.gz")
        p2.write_bytes(gzip.compress(data2, 6))
    if batch:
        monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", str(batch))
    for reader in ("device", "device-text", "host"):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        o1, o2, k = tmp_path / ("o1_" + reader), tmp_path / ("o2_" + reader), tmp_path / ("k_" + reader)
        with Engine.open(DB) as eng:
            st = eng.run(str(p1), str(o1), in2=str(p2) if p2 else None, out2=str(o2) if p2 else None,
                         kraken_output=str(k) if kw.pop("kraken", False) or kw.get("_k") else None, out_codec=out_codec, threads=4,
                         **{a: b for a, b in kw.items() if not a.startswith("_")})
        rd = (lambda p: gzip.decompress(p.read_bytes())) if out_codec == 2 else (lambda p: p.read_bytes())
        outs[reader] = (rd(o1), rd(o2) if p2 else b"", k.read_bytes() if k.exists() else b"",
                        (st.total_sequences, st.classified, st.total_bases))
    monkeypatch.delenv("NOHUMAN_BATCH_FRAGS", raising=False)
    assert outs["device"] == outs["host"], name
    assert outs["device-text"] == outs["host"], name
    return outs["host"]


@pytest.mark.parametrize("out_codec", [0, 2])
def test_record_index_on_the_gpu_keeps_kraken2s_record_semantics(tmp_path, monkeypatch, out_codec):
    """The FASTQ shapes the host parser is tested with (tests/test_reader.py), through the device record index: CRLF, "+id" lines,
    descriptions and trailing blanks on header lines, empty sequences, no newline at the end, a truncated last record
    (dropped), an empty line that ends the input, lower case and N; small batches, so that records are carried from one
    piece and batch to the next; plain and GPU-gzip outputs (the latter take kept records straight from HBM)."""
    rs = read_fastq(os.path.join(GOLD, "reads_se.fq"))
    def rec(i, h=None, seq=None, plus=b"+", nl=b"\n", q=None):
        h0, _id, s0, q0 = rs[i]
        s1 = s0 if seq is None else seq
        return (h if h is not None else h0) + nl + s1 + nl + plus + nl + (q if q is not None else q0[:len(s1)] + b"I" * max(0, len(s1) - len(q0))) + nl
    parts = []
    for i in range(len(rs)):
        kind = i % 9
        if kind == 0: parts.append(rec(i, nl=b"\r\n"))
        elif kind == 1: parts.append(rec(i, plus=b"+" + rs[i][0][1:]))
        elif kind == 2: parts.append(rec(i, h=rs[i][0] + b" some description\t x  "))
        elif kind == 3: parts.append(rec(i, seq=b"", q=b""))
        elif kind == 4: parts.append(rec(i, seq=rs[i][2].lower()))
        else: parts.append(rec(i))
    body = b"".join(parts)
    cases = {
        "plain": body,
        "no_final_newline": body[:-1],
        "truncated_last_record": body + b"@cut\nACGTACGT\n+\n",
        "empty_line_ends_input": body + b"\n" + rec(0) + rec(1),
        "only_three_lines": b"@x\nACGT\n+\n",
        "empty_file": b"",
    }
    for name, data in cases.items():
        for batch in (7, 64):
            got = _run_both_readers(tmp_path, monkeypatch, "%s_%d_%d" % (name, batch, out_codec), data, out_codec=out_codec, batch=batch, _k=True)
            if name == "empty_line_ends_input":
                assert got[3][0] == len(rs)  # what follows the empty line was never read
            if name == "truncated_last_record":
                assert got[3][0] == len(rs)


def test_record_index_on_the_gpu_pairs_and_classified_out(tmp_path, monkeypatch):
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read()
    raw2 = open(os.path.join(GOLD, "reads_pe_2.fq"), "rb").read()
    for keep_human in (False, True):
        for codec in (0, 2):
            _run_both_readers(tmp_path, monkeypatch, "pe_%d_%d" % (keep_human, codec), raw1 * 3, raw2 * 3, out_codec=codec, batch=50,
                              keep_human=keep_human, _k=True)


def test_fasta_and_garbage_fall_back_to_the_host_parser(tmp_path, monkeypatch):
    from nohuman_amd import Engine, EngineError
    rs = read_fastq(os.path.join(GOLD, "reads_se.fq"))
    fa = b"".join(b">" + r[1] + b"\n" + b"".join(r[2][j:j + 60] + b"\n" for j in range(0, len(r[2]), 60)) for r in rs)
    _run_both_readers(tmp_path, monkeypatch, "fasta", fa, _k=True)
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(gzip.compress(b"this is not fastq\n" * 50))
    for reader in ("device", "host"):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        with Engine.open(DB) as eng:
            with pytest.raises(EngineError) as ei:
                eng.run(str(bad), str(tmp_path / "o.fq"))
            assert "unrecognized file format" in str(ei.value)
    mid = tmp_path / "mid.fq.gz"  # a record in the middle that does not start with '@'
    raw = open(os.path.join(GOLD, "reads_se.fq"), "rb").read()
    cut = raw.index(b"\n@", len(raw) // 2) + 1
    mid.write_bytes(gzip.compress(raw[:cut] + b"Xbroken\nACGT\n+\nIIII\n" + raw[cut:]))
    for reader in ("device", "host"):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        with Engine.open(DB) as eng:
            with pytest.raises(EngineError) as ei:
                eng.run(str(mid), str(tmp_path / "o.fq"))
            assert "malformed FASTQ file (exp. '@', saw \"Xbroken\")" in str(ei.value), str(ei.value)


def test_damaged_gzip_input_fails_the_run_with_either_reader(tmp_path, monkeypatch):
    from nohuman_amd import Engine, EngineError
    raw = open(os.path.join(GOLD, "reads_se.fq"), "rb").read() * 20
    good = gzip.compress(raw, 6)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x5A
    p = tmp_path / "bad.fq.gz"
    p.write_bytes(bytes(bad))
    for reader in ("device", "host"):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        with Engine.open(DB) as eng:
            with pytest.raises(EngineError):
                eng.run(str(p), str(tmp_path / "o.fq"), threads=4)


def test_run_errors_are_reported(tmp_path):
    from nohuman_amd import CommandRunner, Engine, EngineError
    with pytest.raises(EngineError) as ei:
        Engine.open(str(tmp_path))
    assert "Required files (hash.k2d, opts.k2d, taxo.k2d) not found" in str(ei.value)
    with Engine.open(DB) as eng:
        with pytest.raises(EngineError):
            eng.run(str(tmp_path / "missing.fq"), str(tmp_path / "o.fq"))
        bad = tmp_path / "bad.fq"
        bad.write_bytes(b"this is not fastq\n")
        with pytest.raises(EngineError) as ei:
            eng.run(str(bad), str(tmp_path / "o.fq"))
        assert "unrecognized file format" in str(ei.value)
    with pytest.raises(OSError) as ei:
        CommandRunner("kraken2").run(["--db", DB, "--unclassified-out", str(tmp_path / "o.fq"),
                                      str(tmp_path / "missing.fq")])
    assert str(ei.value).startswith("kraken2 failed with stderr ")


def test_db_directory_with_db_subdir(tmp_path):
    """<dir>/db/*.k2d is accepted like <dir>/*.k2d (/root/reference/src/lib.rs:119-141)."""
    import shutil
    from nohuman_amd import Engine
    shutil.copytree(DB, tmp_path / "outer" / "db")
    with Engine.open(str(tmp_path / "outer")) as eng:
        assert eng.info.k == 35 and eng.info.l == 31


def test_kraken_report(tmp_path):
    """-r/--kraken-report (/root/reference/src/main.rs:97-99,226-228): kraken-style report lines
    `%6.2f \\t clade \\t taxon \\t rank \\t taxid \\t indented name`, children by descending clade
    count, unclassified line first."""
    from nohuman_amd import Engine
    from oracle import minidb
    from tests import synth
    conf = 0.0
    _, ext, recs, calls = _expected("expected_se.json", conf)
    inp = os.path.join(GOLD, "reads_se.fq")
    rep = tmp_path / "report.txt"
    with Engine.open(DB) as eng:
        eng.run(inp, str(tmp_path / "o.fq"), report=str(rep), confidence=conf)
    tax = minidb.Taxonomy(synth.TOY_EDGES)
    n = tax.node_count
    own = [0] * n
    for c in calls:
        if c:
            own[c] += 1
    clade = own[:]
    for i in range(n - 1, 1, -1):
        clade[tax.parent[i]] += clade[i]
    total = len(calls)
    uncl = sum(1 for c in calls if not c)
    kids = {i: [j for j in range(1, n) if tax.parent[j] == i] for i in range(n)}
    lines = []
    if uncl:
        lines.append("%6.2f\t%d\t%d\tU\t0\tunclassified" % (100.0 * uncl / total, uncl, uncl))

    def dfs(i, depth, rank_depth):
        if clade[i] == 0:
            return
        rank_depth += 1  # every toy rank is "no rank": the code stays R, the depth counts up
        rank = "R" + (str(rank_depth) if rank_depth else "")
        lines.append("%6.2f\t%d\t%d\t%s\t%d\t%s%s" % (100.0 * clade[i] / total, clade[i], own[i], rank,
                                                     tax.external[i], "  " * depth,
                                                     "taxon%d" % tax.external[i]))
        for j in sorted(kids[i], key=lambda j: -clade[j]):
            dfs(j, depth + 1, rank_depth)

    dfs(1, 0, -1)
    assert rep.read_text().split("\n")[:-1] == lines
    assert len(lines) >= 8


def test_many_small_batches_two_engines_keep_input_order(tmp_path, monkeypatch):
    """nh_run with tiny batches (64 fragments) spread round-robin over two engines (both on GPU 0,
    which exercises exactly the multi-device path: one database replica per engine, two stream slots
    each, ordered writer): outputs must equal the single-batch run byte for byte."""
    from nohuman_amd import engine
    _, ext, recs, calls = _expected("expected_pe.json", 0.0)
    in1, in2 = os.path.join(GOLD, "reads_pe_1.fq"), os.path.join(GOLD, "reads_pe_2.fq")
    ref = engine.run(DB, in1, str(tmp_path / "a_1.fq"), in2=in2, out2=str(tmp_path / "a_2.fq"),
                     kraken_output=str(tmp_path / "a.k"), device_ids=[0])
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "64")
    st = engine.run(DB, in1, str(tmp_path / "b_1.fq"), in2=in2, out2=str(tmp_path / "b_2.fq"),
                    kraken_output=str(tmp_path / "b.k"), device_ids=[0, 0])
    assert (st.total_sequences, st.classified, st.total_bases) == (ref.total_sequences, ref.classified,
                                                                     ref.total_bases)
    assert st.total_sequences == len(calls) and st.classified == sum(1 for c in calls if c)
    for n in ("_1.fq", "_2.fq", ".k"):
        assert (tmp_path / ("a" + n)).read_bytes() == (tmp_path / ("b" + n)).read_bytes()


def test_two_engines_on_device_0_sharded(tmp_path, monkeypatch, capfd):
    """one run over G devices -- the pieces of each gzip file's compressed stream go round the run's
    devices (piece i is inflated, indexed and classified on device i mod G; the stream's window moves from device to device,
    the partial batch behind a piece too), the ordered writer is unchanged.  Two engines, both on GPU 0 (the multi-device
    path on one device: two decoder buffer sets, two lanes a file, four more stream slots): the same output bytes,
    kraken lines and counters as one engine, for plain and gzip outputs, and as the host reader."""
    from nohuman_amd import engine
    _, ext, recs, calls = _expected("expected_pe.json", 0.0)
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read()
    raw2 = open(os.path.join(GOLD, "reads_pe_2.fq"), "rb").read()
    # forty copies, pieces of 4 KiB of gzip (a piece ends at a block boundary: some 35 KB each): a dozen pieces a file
    in1, in2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    in1.write_bytes(gzip.compress(raw1 * 40, 6))
    in2.write_bytes(gzip.compress(raw2 * 25, 1) + gzip.compress(raw2 * 15, 9))
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "4096")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "1024")
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "96")
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    res = {}
    for tag, ids, reader, codec in (("one", [0], "device", 0), ("two", [0, 0], "device", 0), ("two_gz", [0, 0], "device", 2),
                                    ("two_host", [0, 0], "host", 0), ("three", [0, 0, 0], "device", 0)):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        o1, o2, k = tmp_path / (tag + "_1"), tmp_path / (tag + "_2"), tmp_path / (tag + ".k")
        st = engine.run(DB, str(in1), str(o1), in2=str(in2), out2=str(o2), kraken_output=str(k), device_ids=ids, out_codec=codec,
                        threads=4)
        err = capfd.readouterr().err
        rd = (lambda p: gzip.decompress(p.read_bytes())) if codec == 2 else (lambda p: p.read_bytes())
        res[tag] = (rd(o1), rd(o2), k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))
        if reader == "device" and len(ids) > 1:
            # both lanes of each file decoded pieces: "pieces by device (device:pieces) 0:a 0:b" with a, b > 0
            lines = [ln for ln in err.splitlines() if "pieces by device" in ln]
            assert len(lines) == 2, err[-3000:]
            for ln in lines:
                counts = [int(x.split(":")[1]) for x in ln.split(")")[-1].split()]
                assert len(counts) == len(ids) and min(counts) >= 2, ln
    n = len(calls) * 40
    assert res["one"][3][0] == n and res["one"][3][1] == 40 * sum(1 for c in calls if c)
    for tag in ("two", "two_gz", "two_host", "three"):
        assert res[tag] == res["one"], tag


def test_pieces_decoded_ahead_on_several_lanes(tmp_path, monkeypatch, capfd):
    """The reader over several devices decodes the pieces of the grid AHEAD of the stream, a lane each at the same time
    (search + decode need neither the window nor the stream's state), and takes them in stream order.  Three engines on
    GPU 0; cells of 256 KiB of gzip (most pieces decoded ahead are taken), of 16 KiB (a deflate block spans cells: most are
    refused or stale, and the stream decodes in order to the next cell boundary), and a planted false first start
    (NOHUMAN_GZDEV_FAKE_SPEC): always the bytes and counters of one engine."""
    from nohuman_amd import engine
    _, ext, recs, calls = _expected("expected_pe.json", 0.0)
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read()
    raw2 = open(os.path.join(GOLD, "reads_pe_2.fq"), "rb").read()
    in1, in2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    in1.write_bytes(gzip.compress(raw1 * 160, 6))
    in2.write_bytes(gzip.compress(raw2 * 100, 1) + gzip.compress(raw2 * 60, 9))
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "2048")
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    monkeypatch.setenv("NOHUMAN_GZ_READER", "device")
    res = {}
    for tag, ids, seg, fake in (("one", [0], 262144, None), ("three", [0, 0, 0], 262144, None), ("three_fake", [0, 0, 0], 262144, "2"),
                                ("two_tiny", [0, 0], 16384, None), ("three_off", [0, 0, 0], 262144, "off")):
        monkeypatch.setenv("NOHUMAN_GZDEV_SEG", str(seg))
        monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "4096")
        monkeypatch.delenv("NOHUMAN_GZDEV_FAKE_SPEC", raising=False)
        monkeypatch.delenv("NOHUMAN_GZ_AHEAD", raising=False)
        if fake == "off":
            monkeypatch.setenv("NOHUMAN_GZ_AHEAD", "0")
        elif fake:
            monkeypatch.setenv("NOHUMAN_GZDEV_FAKE_SPEC", fake)
        o1, o2, k = tmp_path / (tag + "_1"), tmp_path / (tag + "_2"), tmp_path / (tag + ".k")
        st = engine.run(DB, str(in1), str(o1), in2=str(in2), out2=str(o2), kraken_output=str(k), device_ids=ids, threads=4)
        err = capfd.readouterr().err
        res[tag] = (o1.read_bytes(), o2.read_bytes(), k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))
        took = [ln for ln in err.splitlines() if "decoded ahead of the stream taken" in ln]
        if tag in ("three", "three_fake"):
            assert len(took) == 2, err[-3000:]
            for ln in took:
                a, b = int(ln.split(":")[-1].split()[0]), int(ln.split("taken,")[1].split()[0])
                assert a >= 3 and (b >= 1 if fake else True), ln  # pieces were taken (and with the planted start: one refused)
        if tag in ("one", "three_off"):
            assert not took, err[-2000:]
    assert res["one"][3][0] == len(calls) * 160
    for tag in ("three", "three_fake", "two_tiny", "three_off"):
        assert res[tag] == res["one"], tag


def test_lanes_with_pieces_the_host_decoder_takes_over(tmp_path, monkeypatch, capfd):
    """Reads of a million identical bases (text beyond 16 : 1 in a chunk) between ordinary ones: those pieces are decoded by the
    host decoder -- loudly --, the stream leaves the piece grid and finds back to it; with three lanes decoding ahead the
    outputs are one engine's and the host reader's."""
    from nohuman_amd import engine
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read()
    big = b"@long.%d\n" + b"A" * 1_000_000 + b"\n+\n" + b"I" * 1_000_000 + b"\n"
    data = raw1 * 30 + big % 1 + raw1 * 25 + big % 2 + big % 3 + raw1 * 40
    in1 = tmp_path / "r.fq.gz"
    in1.write_bytes(gzip.compress(data, 6))
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "2048")
    monkeypatch.setenv("NOHUMAN_GZDEV_SEG", "262144")
    monkeypatch.setenv("NOHUMAN_GZDEV_STRETCH", "4096")
    res = {}
    for tag, ids, reader in (("one", [0], "device"), ("three", [0, 0, 0], "device"), ("host", [0], "host")):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        o1, k = tmp_path / (tag + "_1"), tmp_path / (tag + ".k")
        st = engine.run(DB, str(in1), str(o1), kraken_output=str(k), device_ids=ids, threads=4)
        err = capfd.readouterr().err
        res[tag] = (o1.read_bytes(), k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))
        if reader == "device":
            assert "decoded on the host" in err, err[-1500:]
    assert res["one"][2][0] == 95 * (len(raw1.split(b"\n")) // 4) + 3
    assert res["three"] == res["one"] and res["host"] == res["one"]


def test_bgzf_inputs_through_the_reader_on_the_gpu(tmp_path, monkeypatch, capfd):
    """bgzip-compressed FASTQ pairs through nh_run: the device reader takes BGZF (chunk starts from the members' headers), the
    outputs are the host reader's."""
    from nohuman_amd import Engine
    from tests.test_gpu_gunzip import bgzf
    raw1 = open(os.path.join(GOLD, "reads_pe_1.fq"), "rb").read() * 12
    raw2 = open(os.path.join(GOLD, "reads_pe_2.fq"), "rb").read() * 12
    in1, in2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    in1.write_bytes(bgzf(raw1))
    in2.write_bytes(bgzf(raw2, block=20000, level=1))
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "1000")
    res = {}
    for reader in ("device", "host"):
        monkeypatch.setenv("NOHUMAN_GZ_READER", reader)
        o1, o2, k = tmp_path / ("o1_" + reader), tmp_path / ("o2_" + reader), tmp_path / ("k_" + reader)
        with Engine.open(DB) as eng:
            st = eng.run(str(in1), str(o1), in2=str(in2), out2=str(o2), kraken_output=str(k), threads=4)
        err = capfd.readouterr().err
        assert ("gzip reader on GPU" in err) == (reader == "device"), err[-1500:]
        if reader == "device":
            assert "0 pieces by the host decoder" in err
        res[reader] = (o1.read_bytes(), o2.read_bytes(), k.read_bytes(), (st.total_sequences, st.classified, st.total_bases))
    assert res["device"] == res["host"] and res["host"][3][0] == 12 * (raw1.count(b"\n") // 48)


def test_fragments_with_many_taxa_in_concurrent_batches(tmp_path, monkeypatch):
    """Batches in flight on the two stream slots of an engine each carry fragments that hit more than
    64 distinct taxa (second kernel pass): every launch has its own 'left for the second pass' word,
    so no batch may lose such a fragment.  Compared per read with the oracle."""
    import numpy as np
    from nohuman_amd import engine
    from oracle import minidb
    from oracle import oracle as orc
    from tests import synth
    rng = np.random.default_rng(33)
    edges = {1: 0}
    for g in range(10):
        edges[100 + g] = 1
    leaves = []
    for i in range(150):
        edges[1000 + i] = 100 + i % 10
        leaves.append(1000 + i)
    tax = minidb.Taxonomy(edges)
    segs = {e: synth.random_seq(rng, 120) for e in leaves}
    hashb, _ = minidb.build_hash(tax, sorted(segs.items()), 40009)
    db = tmp_path / "db"
    db.mkdir()
    (db / "opts.k2d").write_bytes(minidb.opts_bytes())
    (db / "taxo.k2d").write_bytes(tax.to_bytes())
    (db / "hash.k2d").write_bytes(hashb)
    reads = []
    for i in range(600):
        if i % 3 == 0:  # every third read visits 70-140 leaves
            pick = rng.choice(leaves, size=int(rng.integers(70, 141)), replace=False)
            reads.append(b"".join(segs[int(e)][10:110] for e in pick))
        else:
            reads.append(synth.random_seq(rng, 150) if i % 3 == 1 else segs[leaves[i % 150]][5:115])
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads)))
    bases, offs = orc.pack_reads(reads, False)
    odb = orc.OracleDB(minidb.opts_bytes(), tax.to_bytes(), hashb)
    exp, _ = odb.classify(bases, offs, False, 0.0)
    monkeypatch.setenv("NOHUMAN_BATCH_FRAGS", "16")  # 38 batches over 2 slots
    st = engine.run(str(db), str(fq), str(tmp_path / "o.fq"), kraken_output=str(tmp_path / "o.k"), device_ids=[0])
    lines = (tmp_path / "o.k").read_text().splitlines()
    assert len(lines) == len(reads)
    got_ext = np.array([int(l.split("\t")[2]) for l in lines])
    want_ext = np.asarray(odb.external_ids)[exp["call"]].astype(np.int64)
    assert np.array_equal(got_ext, want_ext)
    assert st.classified == int((exp["call"] != 0).sum())


def test_ultra_long_reads_gzip_fasta_and_fastq(tmp_path, toy, toy_oracle, monkeypatch):
    """Reads of 0.3-1.5 Mb (records far longer than the 4 MiB read granularity is not, but longer than
    one gzip chunk): gzip FASTQ through the multi-threaded decoder and 60-column FASTA (joined in place),
    compared per read with the oracle; the kept records must come back byte for byte."""
    import numpy as np
    from nohuman_amd import engine
    from oracle import oracle as orc
    from tests import synth
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(42)
    allg = b"".join(genomes[k] for k in sorted(genomes))
    reads = []
    for ln in (1_500_000, 300_000, 777_777, 35, 20, 1_000_001, 150):
        parts = []
        while sum(map(len, parts)) < ln:
            if rng.random() < 0.5:
                st = int(rng.integers(0, len(allg) - 2000))
                parts.append(allg[st:st + int(rng.integers(200, 2000))])
            else:
                parts.append(synth.random_seq(rng, int(rng.integers(500, 5000))))
        reads.append(synth.mutate(rng, b"".join(parts)[:ln], 0.03, 0.0005, 0.0))
    bases, offs = orc.pack_reads(reads, False)
    exp, _ = toy_oracle.classify(bases, offs, False, 0.0)
    want_ext = np.asarray(toy_oracle.external_ids)[exp["call"]].astype(np.int64)
    monkeypatch.setenv("NOHUMAN_GZ_CHUNK", "65536")
    fq = b"".join(b"@long%d len=%d\n%s\n+\n%s\n" % (i, len(r), r, b"5" * len(r)) for i, r in enumerate(reads))
    fa = b"".join(b">long%d\n" % i + b"".join(r[j:j + 60] + b"\n" for j in range(0, len(r), 60)) for i, r in enumerate(reads))
    for name, data in (("r.fq.gz", fq), ("r.fa.gz", fa)):
        p = tmp_path / name
        p.write_bytes(gzip.compress(data, 6))
        out, k = tmp_path / (name + ".out"), tmp_path / (name + ".k")
        st = engine.run(DB, str(p), str(out), kraken_output=str(k), threads=4, device_ids=[0])
        lines = k.read_text().splitlines()
        assert [int(l.split("\t")[2]) for l in lines] == list(want_ext), name
        assert [int(l.split("\t")[3]) for l in lines] == [len(r) for r in reads]
        assert st.total_bases == sum(len(r) for r in reads)
        kept = [i for i, c in enumerate(exp["call"]) if c == 0]
        if name.endswith("fq.gz"):
            want = b"".join(b"@long%d len=%d\n%s\n+\n%s\n" % (i, len(reads[i]), reads[i], b"5" * len(reads[i])) for i in kept)
        else:
            want = b"".join(b">long%d\n%s\n" % (i, reads[i]) for i in kept)
        assert out.read_bytes() == want, name
    assert (exp["call"] != 0).sum() >= 3


def test_count_allreduce_over_rccl_single_device(tmp_path, monkeypatch, capfd):
    """the run's only collective is ONE ncclAllReduce(4 x uint64, sum) over the per-device
    counters (RCCL, single process, ncclCommInitAll).  On a 1-GPU box the communicator has one rank: the
    row must come back unchanged, and nh_run with NOHUMAN_RCCL=1 must agree with its host-side sum."""
    import ctypes as C
    from nohuman_amd import Engine, _lib
    L = _lib.lib()
    rows = (C.c_uint64 * 4)(1001, 250, 150150, 39039)
    backend = C.create_string_buffer(512)
    rc = L.nh_allreduce_counters(None, 1, rows, backend, 512)
    assert rc == 0, backend.value
    assert list(rows) == [1001, 250, 150150, 39039]
    assert b"RCCL" in backend.value and b"ncclAllReduce" in backend.value
    print("collective:", backend.value.decode())
    # the product path: NOHUMAN_RCCL=1 runs the collective even for one device and checks it
    monkeypatch.setenv("NOHUMAN_RCCL", "1")
    monkeypatch.setenv("NOHUMAN_TRACE", "1")
    _, _, _, calls = _expected("expected_se.json", 0.0)
    with Engine.open(DB) as eng:
        st = eng.run(os.path.join(GOLD, "reads_se.fq"), str(tmp_path / "o.fq"))
        folded = eng.stats()
    assert st.classified == sum(1 for c in calls if c)
    # the run's own device-resident counters were folded into the engine's running totals
    assert (folded.total_sequences, folded.classified) == (st.total_sequences, st.classified)
    assert folded.table_lookups > 0
    err = capfd.readouterr().err
    assert "counters reduced by RCCL" in err and "device-resident counters" in err


def test_count_allreduce_failure_is_loud(tmp_path, monkeypatch, capfd):
    """at G > 1 a collective that cannot run must not pass silently.  Two engines on the SAME
    device make ncclCommInitAll fail (duplicate device): by default the run finishes on the host-side sums
    with one WARN line on stderr; NOHUMAN_RCCL=strict fails the run; NOHUMAN_RCCL=0 skips the collective."""
    from nohuman_amd import engine
    in1 = os.path.join(GOLD, "reads_se.fq")
    monkeypatch.delenv("NOHUMAN_RCCL", raising=False)
    st = engine.run(DB, in1, str(tmp_path / "a.fq"), device_ids=[0, 0])
    err = capfd.readouterr().err
    assert "WARN count all-reduce over 2 devices did not run" in err
    monkeypatch.setenv("NOHUMAN_RCCL", "0")
    st0 = engine.run(DB, in1, str(tmp_path / "b.fq"), device_ids=[0, 0])
    assert "WARN" not in capfd.readouterr().err
    assert (st0.total_sequences, st0.classified) == (st.total_sequences, st.classified)
    assert (tmp_path / "a.fq").read_bytes() == (tmp_path / "b.fq").read_bytes()
    monkeypatch.setenv("NOHUMAN_RCCL", "strict")
    with pytest.raises(RuntimeError) as ei:
        engine.run(DB, in1, str(tmp_path / "c.fq"), device_ids=[0, 0])
    assert "ncclCommInitAll" in str(ei.value)


@pytest.mark.parametrize("codec", [1, 2, 3, 4])
def test_run_streams_kept_pairs_into_the_output_codec(tmp_path, codec):
    """nh_run_args.out_codec (ABI 2): the writer's spans go through a streaming encoder; what comes out
    decompresses to exactly the plain-text outputs."""
    import bz2
    import lzma
    from nohuman_amd import Engine
    from tests.test_codec import _zstd_decompress
    ins = [os.path.join(GOLD, "reads_pe_1.fq"), os.path.join(GOLD, "reads_pe_2.fq")]
    with Engine.open(DB) as eng:
        eng.run(ins[0], str(tmp_path / "p_1.fq"), in2=ins[1], out2=str(tmp_path / "p_2.fq"), confidence=0.1)
        eng.run(ins[0], str(tmp_path / "c_1"), in2=ins[1], out2=str(tmp_path / "c_2"), confidence=0.1,
                out_codec=codec, codec_threads=3)
    for m in ("1", "2"):
        want = (tmp_path / ("p_%s.fq" % m)).read_bytes()
        raw = (tmp_path / ("c_" + m)).read_bytes()
        got = {1: bz2.decompress, 2: gzip.decompress, 3: lzma.decompress,
               4: lambda b: _zstd_decompress(b, len(want))}[codec](raw)
        assert got == want and len(want) > 1000


def test_run_refuses_an_output_that_is_an_input(tmp_path):
    """outputs are created with O_TRUNC before a byte of input is read; nh_run must refuse an
    output path that names an input (same inode) instead of emptying it."""
    import shutil
    from nohuman_amd import engine
    src = tmp_path / "reads.fq"
    shutil.copy(os.path.join(GOLD, "reads_se.fq"), src)
    size = src.stat().st_size
    os.link(src, tmp_path / "alias.fq")
    for out in (src, tmp_path / "alias.fq"):
        with pytest.raises(RuntimeError) as ei:
            engine.run(DB, str(src), str(out), device_ids=[0])
        assert "is the input" in str(ei.value)
    assert src.stat().st_size == size


def test_run_with_pageable_batch_buffers(tmp_path):
    """a host that will not page-lock the batch text buffers (memlock / cgroup limits) must not fail
    the run: the buffers fall back to pageable memory (NOHUMAN_NO_PINNED=1 forces it; the switch is read once per
    process, hence the child) and the outputs stay byte-identical."""
    import subprocess
    import sys
    from nohuman_amd import engine
    in1, in2 = os.path.join(GOLD, "reads_pe_1.fq"), os.path.join(GOLD, "reads_pe_2.fq")
    engine.run(DB, in1, str(tmp_path / "a_1.fq"), in2=in2, out2=str(tmp_path / "a_2.fq"), device_ids=[0])
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from nohuman_amd import engine\n"
            "st = engine.run(%r, %r, sys.argv[1], in2=%r, out2=sys.argv[2], device_ids=[0])\n"
            "print(st.total_sequences, st.classified)\n") % (ROOT, DB, in1, in2)
    env = dict(os.environ, NOHUMAN_NO_PINNED="1", NOHUMAN_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "b_1.fq"), str(tmp_path / "b_2.fq")], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "could not be page-locked" in r.stderr
    for n in ("_1.fq", "_2.fq"):
        assert (tmp_path / ("a" + n)).read_bytes() == (tmp_path / ("b" + n)).read_bytes()
