"""§8 row a4: hash.k2d goes to HBM through several loaders that take the file's 64 MiB chunks in turn (round 5).  Whatever the number
of loaders and however the file's size falls on the chunk grid, the table in HBM is the file, cell for cell."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("loaders", ["1", "3", "4", "16"])
def test_table_in_hbm_equals_the_file_for_any_number_of_loaders(tmp_path, monkeypatch, loaders):
    from nohuman_amd import Engine
    cap = 58_720_301  # 224 MiB of cells + 44 bytes: three whole chunks and a ragged fourth
    with Engine.synthetic(cap, int(cap * 0.6), depth=12, seed=5) as eng:
        info = eng.info
        cells = eng.download_table()
        db = tmp_path / "db"
        db.mkdir()
        (db / "opts.k2d").write_bytes(eng.opts_image())
        (db / "taxo.k2d").write_bytes(eng.taxonomy_image())
        with open(db / "hash.k2d", "wb") as f:
            f.write(struct.pack("<4Q", info.capacity, info.size, info.key_bits, info.value_bits))
            cells.tofile(f)
    monkeypatch.setenv("NOHUMAN_DB_LOADERS", loaders)
    with Engine.open(str(db)) as eng2:
        got = eng2.download_table()
        assert eng2.info.capacity == cap and eng2.info.size == info.size
    assert np.array_equal(got, cells)
    assert int(np.count_nonzero(got)) == info.size  # the format's own equation
