"""Real-format evidence without a kraken2 binary: when NOHUMAN_PIN_DB names a directory with
a REAL hash.k2d / opts.k2d / taxo.k2d (e.g. HPRC.r2, /root/reference/config.toml:1-7), the engine must load it
as it is, its headers must satisfy the format's own equations, and reads classified on the GPU
must equal the CPU oracle's on the same files.  Skips with the reason printed where no such directory exists
(this pool: no network, no database) -- the status tables keep saying UNPINNED until it has run."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PIN = os.environ.get("NOHUMAN_PIN_DB")


@pytest.mark.skipif(not PIN, reason="NOHUMAN_PIN_DB is not set: no real kraken2 database on this box")
def test_real_database_loads_and_agrees_with_the_oracle():
    from nohuman_amd import Engine, validate_db_directory
    from oracle import oracle as orc
    from tests import synth
    d = validate_db_directory(PIN)
    sizes = {n: os.path.getsize(os.path.join(d, n)) for n in ("hash.k2d", "opts.k2d", "taxo.k2d")}
    with Engine.open(d) as eng:
        i = eng.info
        print("real database %s: capacity %d, size %d (load %.3f), key_bits %d, value_bits %d, nodes %d, k %d, l %d, "
              "spaced 0x%x, toggle 0x%x, min_hash %d, revcom_version %d"
              % (d, i.capacity, i.size, i.size / i.capacity, i.key_bits, i.value_bits, i.node_count, i.k, i.l,
                 i.spaced_seed_mask, i.toggle_mask, i.minimum_acceptable_hash_value, i.revcom_version))
        # the format's own equations
        assert sizes["hash.k2d"] == 32 + 4 * i.capacity
        assert i.key_bits + i.value_bits == 32 and (1 << i.value_bits) >= i.node_count
        assert i.l <= i.k and i.l <= 31 and bin(i.spaced_seed_mask).count("1") % 2 == 0
        assert 0 < i.size < i.capacity
        cells = eng.download_table()
        assert int(np.count_nonzero(cells & np.uint32((1 << i.value_bits) - 1))) == i.size  # occupied cells == header size
        del cells
        odb = orc.OracleDB(directory=str(d))  # the oracle reads the three files with its own parser
        rng = np.random.default_rng(1)
        reads = [synth.random_seq(rng, 150) for _ in range(20_000)]
        human = os.environ.get("NOHUMAN_BENCH_HUMAN_FASTA")
        if human and os.path.exists(human):  # hit-bearing reads: what pins LINEAR_PROBING
            seq = b"".join(l.strip() for l in open(human, "rb").readlines()[1:20000]).upper()
            for _ in range(20_000):
                st = int(rng.integers(0, max(1, len(seq) - 150)))
                reads.append(synth.mutate(rng, seq[st:st + 150], 0.01, 0.0, 0.0))
        for paired in (False, True):
            rs = reads if not paired else list(zip(reads[0::2], reads[1::2]))
            bases, offs = orc.pack_reads(rs, paired)
            exp, lookups = odb.classify(bases, offs, paired, 0.0)
            got = eng.classify(bases, offs, paired, 0.0)
            for f in ("call", "total_kmers", "clade_hits", "hit_groups"):
                assert np.array_equal(got[f], exp[f]), f
