"""§8 row a4 / (b): nh_open* checks the CONTENT of the database, not only its sizes (round 6).

The kernels index the taxonomy with a cell's value unchecked; the reference only checks that the three files exist
(/root/reference/src/lib.rs:119-141) and several database versions can be installed side by side
(/root/reference/src/download.rs:178-222), so a hash.k2d beside another database's taxo.k2d is a real way to get here.
The header promises "never aborts across the ABI": each case below must come back as NH_EDB (-3) with both numbers in
the message, and the process -- this test process -- must still be alive and able to classify afterwards."""
import struct

import numpy as np
import pytest

from oracle import minidb
from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu

NH_EDB = -3


def _cells(hashb):
    cap, size, kb, vb = struct.unpack_from("<4Q", hashb, 0)
    return cap, size, kb, vb, np.frombuffer(hashb, dtype="<u4", offset=32).copy()


def _image(cap, size, kb, vb, cells):
    return struct.pack("<4Q", cap, size, kb, vb) + cells.astype("<u4").tobytes()


def _still_works(toy, toy_oracle):
    from nohuman_amd import Engine
    ob, tb, hb, genomes, _ = toy
    reads = synth.sample_reads(np.random.default_rng(3), genomes, 64)
    bases, offs = orc.pack_reads(reads, False)
    exp, _ = toy_oracle.classify(bases, offs, False, 0.0)
    with Engine.from_images(ob, tb, hb) as eng:
        got = eng.classify(bases, offs, False, 0.0)
        chk = eng.db_check()
        info = eng.info
    assert np.array_equal(got["call"], exp["call"])
    return chk, info


def test_a_good_database_reports_what_the_check_measured(toy, toy_oracle):
    _, _, hb, _, _ = toy
    cap, size, _, vb, cells = _cells(hb)
    chk, info = _still_works(toy, toy_oracle)
    assert chk.non_empty_cells == size == info.size == int(np.count_nonzero(cells & ((1 << vb) - 1)))
    assert chk.max_value == int((cells & ((1 << vb) - 1)).max()) < info.node_count
    assert abs(chk.load_factor - size / cap) < 1e-12


def test_a_cell_value_of_node_count_is_refused_not_dereferenced(toy, toy_oracle):
    from nohuman_amd import Engine, EngineError
    ob, tb, hb, _, _ = toy
    cap, size, kb, vb, cells = _cells(hb)
    node_count = struct.unpack_from("<Q", tb, 8)[0]
    assert node_count < (1 << vb)  # the value fits the cell: only the taxonomy says it is too large
    used = np.flatnonzero(cells)
    for where in (used[0], used[len(used) // 2], used[-1]):
        bad = cells.copy()
        bad[where] = (bad[where] >> vb << vb) | node_count
        with pytest.raises(EngineError) as ei:
            Engine.from_images(ob, tb, _image(cap, size, kb, vb, bad))
        assert ei.value.code == NH_EDB
        assert str(node_count) in ei.value.message and "taxo.k2d" in ei.value.message
    _still_works(toy, toy_oracle)


def test_hash_beside_a_smaller_taxonomy_is_refused(toy, toy_oracle, tmp_path):
    """Two installed versions mixed up: the directory form of the same mistake (nh_open)."""
    from nohuman_amd import Engine, EngineError
    ob, tb, hb, _, _ = toy
    small = minidb.Taxonomy({1: 0, 10: 1, 20: 1}).to_bytes()  # 4 nodes; the toy table holds values up to 9
    with pytest.raises(EngineError) as ei:
        Engine.from_images(ob, small, hb)
    assert ei.value.code == NH_EDB and "4 nodes" in ei.value.message
    db = tmp_path / "mixed" / "db"
    db.mkdir(parents=True)
    (db / "opts.k2d").write_bytes(ob)
    (db / "taxo.k2d").write_bytes(small)
    (db / "hash.k2d").write_bytes(hb)
    with pytest.raises(EngineError) as ei:
        Engine.open(str(tmp_path / "mixed"))
    assert ei.value.code == NH_EDB and "not of one database" in ei.value.message
    _still_works(toy, toy_oracle)


def test_header_size_must_be_the_number_of_cells_in_use(toy, toy_oracle):
    from nohuman_amd import Engine, EngineError
    ob, tb, hb, _, _ = toy
    cap, size, kb, vb, cells = _cells(hb)
    for claimed in (size + 1, size - 1, 0):
        with pytest.raises(EngineError) as ei:
            Engine.from_images(ob, tb, _image(cap, claimed, kb, vb, cells))
        assert ei.value.code == NH_EDB
        assert str(size) in ei.value.message and str(claimed) in ei.value.message
    emptied = cells.copy()
    emptied[np.flatnonzero(cells)[5]] = 0  # a cell lost: the header no longer describes the table
    with pytest.raises(EngineError) as ei:
        Engine.from_images(ob, tb, _image(cap, size, kb, vb, emptied))
    assert ei.value.code == NH_EDB
    _still_works(toy, toy_oracle)


def test_a_taxonomy_that_loops_is_refused(toy, toy_oracle):
    """The kernels climb with `while (b > a) b = parent[b]`: a root that names a child as its parent, or any parent id
    not below its child, would spin a wave for ever."""
    from nohuman_amd import Engine, EngineError
    ob, tb, hb, _, _ = toy
    for node, parent in ((1, 3), (1, 1), (4, 4), (5, 9)):
        bad = bytearray(tb)
        struct.pack_into("<Q", bad, 32 + 56 * node, parent)
        with pytest.raises(EngineError) as ei:
            Engine.from_images(ob, bytes(bad), hb)
        assert ei.value.code == NH_EDB and "taxo.k2d" in ei.value.message
    _still_works(toy, toy_oracle)


def test_what_the_report_writer_walks_is_checked_too(toy, toy_oracle, tmp_path):
    """-r / --kraken-report (nh_run.hip write_report) walks the children's ranges and reads names and ranks as C strings out of
    taxo.k2d on the HOST: a range that leaves the node table or an offset that leaves its string table is NH_EDB at nh_open, not
    a wild read when the report is written"""
    from nohuman_amd import Engine, EngineError
    ob, tb, hb, _, _ = toy
    nc, nl, rl = struct.unpack_from("<3Q", tb, 8)
    cases = []
    for node, field, value in ((1, 1, nc), (1, 2, nc + 5), (2, 1, 1), (3, 3, nl), (3, 3, nl + 1000), (4, 4, rl), (4, 4, 1 << 40)):
        bad = bytearray(tb)
        if field == 2:
            struct.pack_into("<Q", bad, 32 + 56 * node + 8, node + 1)  # a first child that exists ...
        if field == 1 and value == 1:
            struct.pack_into("<Q", bad, 32 + 56 * node + 16, 1)        # ... / a count, so that the range is looked at
        struct.pack_into("<Q", bad, 32 + 56 * node + 8 * field, value)
        cases.append(bytes(bad))
    unterminated = bytearray(tb)
    unterminated[32 + 56 * nc + nl - 1] = ord("x")
    cases.append(bytes(unterminated))
    for bad in cases:
        with pytest.raises(EngineError) as ei:
            Engine.from_images(ob, bad, hb)
        assert ei.value.code == NH_EDB and "taxo.k2d" in ei.value.message
    # and the good one still writes its report
    reads = tmp_path / "r.fq"
    _, _, _, genomes, _ = toy
    g = genomes[sorted(genomes)[0]]
    reads.write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, g[i * 7:i * 7 + 150], b"I" * 150) for i in range(40)))
    db = tmp_path / "db"
    db.mkdir()
    (db / "opts.k2d").write_bytes(ob)
    (db / "taxo.k2d").write_bytes(tb)
    (db / "hash.k2d").write_bytes(hb)
    with Engine.open(str(db)) as eng:
        eng.run(str(reads), str(tmp_path / "o.fq"), report=str(tmp_path / "rep.txt"))
    rep = (tmp_path / "rep.txt").read_text()
    assert rep.count("\n") >= 1 and "taxon1" in rep  # (reads cut from the segment every genome shares: the root's clade)


def test_the_check_runs_at_hbm_speed_on_a_full_size_table():
    """1.43 G cells = 5.7 GB (BASELINE.json's table size): the pass is a streaming read, a few milliseconds."""
    from nohuman_amd import Engine
    cap = 1_431_655_765
    with Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=1) as eng:
        chk, info = eng.db_check(), eng.info
    assert chk.non_empty_cells == info.size and chk.max_value == 30 < info.node_count
    assert 0.69 < chk.load_factor < 0.70001
    assert chk.seconds < 0.05, "table check took %.4f s" % chk.seconds
