"""BASELINE.json's full sizes on the GPU (configs[1]: 1 M x 150 bp single-end, the bench workload:
1 M x 150 bp pairs, an HPRC.r2-sized table of 1.43 G cells resident in HBM), checked
  * bit-exact against the CPU oracle over the WHOLE batch (the oracle probes the very table that sits
    in HBM, downloaded once), and
  * through properties that do not depend on the size: splitting the batch, permuting the fragments,
    reverse-complementing every read, and repeating the launch leave the records unchanged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAP = 1_431_655_765  # bench.py's HPRC.r2-like table: 5.73 GB
N = 1_000_000
L = 150


class Counters:
    def __init__(self, fragments, classified, bases, lookups):
        self.total_sequences, self.classified, self.total_bases, self.table_lookups = fragments, classified, bases, lookups


@pytest.fixture(scope="module")
def big():
    import torch
    from nohuman_amd import Engine
    dev = torch.device("cuda:0")
    eng = Engine.synthetic(CAP, int(CAP * 0.7), depth=30, seed=20250101)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    codes = torch.randint(0, 4, (2 * N * L,), generator=g, device=dev)
    # a tenth of the reads are cut from sequences put into the table, so hits, LCAs and the
    # confidence climb take part; 0.1 % of all bases are N
    n_hit = N // 10
    bases = acgt[codes]
    offs2 = (torch.arange(2 * N + 1, dtype=torch.int64, device=dev) * L).contiguous()
    pad = torch.full((64,), 65, dtype=torch.uint8, device=dev)
    eng.add_sequences(torch.cat([bases, pad]).contiguous().data_ptr(), offs2.data_ptr(), 2 * n_hit, 30)
    eng.add_sequences(torch.cat([bases[2 * n_hit * L:], pad]).contiguous().data_ptr(), offs2.data_ptr(), n_hit, 17)
    nmask = torch.rand(bases.shape, generator=g, device=dev) < 0.001
    bases = torch.where(nmask, torch.tensor(78, dtype=torch.uint8, device=dev), bases)
    bases = torch.cat([bases, pad]).contiguous()
    torch.cuda.synchronize()
    yield dict(torch=torch, dev=dev, eng=eng, bases=bases, offs2=offs2, acgt=acgt)
    eng.close()


def _classify(b, bases, offs, n, paired, conf=0.0, counters=False):
    torch = b["torch"]
    out = torch.zeros((n, 4), dtype=torch.int32, device=b["dev"])
    cnt = torch.zeros(4, dtype=torch.int64, device=b["dev"])  # fragments, classified, bases, lookups
    b["eng"].classify_device(bases.data_ptr(), offs.data_ptr(), n, paired, conf, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    return (out, Counters(*cnt.tolist())) if counters else out


@pytest.fixture(scope="module")
def oracle_db(big):
    from oracle import oracle as orc
    eng = big["eng"]
    info = eng.info
    cells = eng.download_table()
    return orc.OracleDB(eng.opts_image(), eng.taxonomy_image(), cells=cells,
                        header=(info.capacity, info.size, info.key_bits, info.value_bits))


@pytest.fixture(scope="module")
def oracle_bases_per_second(big, oracle_db):
    """What the CPU oracle manages on THIS box (all usable cores), measured on 100 k reads of the fixture.
    The oracle legs below compare the whole batch wherever that takes less than about 40 s of oracle time
    (any box with a handful of cores); on a slower box they compare the longest prefix that does, so that
    the suite's run time does not depend on the box (980 s of a 1200 s limit on the driver's)."""
    import time
    from nohuman_amd.dist import usable_cpu_count
    n = 100_000
    offs = big["offs2"][: n + 1].cpu().numpy().astype(np.uint64)
    host = big["bases"][: n * L].cpu().numpy()
    oracle_db.classify(host[: 1000 * L], offs[:1001], False, 0.0, threads=usable_cpu_count())  # warm
    t0 = time.perf_counter()
    oracle_db.classify(host, offs, False, 0.0, threads=usable_cpu_count())
    return n * L / (time.perf_counter() - t0)


ORACLE_SECONDS = 40.0


@pytest.mark.parametrize("paired,conf", [(False, 0.0), (True, 0.0), (True, 0.1)])
def test_whole_batch_equals_the_oracle(big, oracle_db, paired, conf):
    from nohuman_amd.dist import usable_cpu_count
    torch = big["torch"]
    mates = 2 if paired else 1
    offs = big["offs2"][: N * mates + 1].contiguous()
    got, st = _classify(big, big["bases"], offs, N, paired, conf, counters=True)
    host = big["bases"][: N * mates * L].cpu().numpy()
    exp, lookups = oracle_db.classify(host, offs.cpu().numpy().astype(np.uint64), paired, conf,
                                      threads=usable_cpu_count())
    rec = got.cpu().numpy().view(np.uint32)
    for i, f in enumerate(("call", "total_kmers", "clade_hits", "hit_groups")):
        bad = np.nonzero(rec[:, i] != exp[f])[0]
        assert bad.size == 0, "%s differs at %s" % (f, bad[:5])
    assert st.total_sequences == N and st.total_bases == N * mates * L
    assert st.classified == int((exp["call"] != 0).sum())
    assert st.table_lookups == int(lookups.sum())
    assert 0.05 * N < st.classified < 0.4 * N  # the hit paths took part


def test_split_permute_repeat_leave_the_records_unchanged(big):
    torch = big["torch"]
    offs = big["offs2"]
    ref, st = _classify(big, big["bases"], offs, N, True, counters=True)
    # repeat: dynamic scheduling must not show in the results
    assert torch.equal(ref, _classify(big, big["bases"], offs, N, True))
    # split into 7 uneven pieces: records concatenate, counters add up
    cuts = [0, 1, 1000, 123_457, 500_000, 500_001, 999_999, N]
    parts, seqs, cls, looks = [], 0, 0, 0
    for a, c in zip(cuts[:-1], cuts[1:]):
        o = offs[2 * a: 2 * c + 1].contiguous()
        out, s = _classify(big, big["bases"], o, c - a, True, counters=True)
        parts.append(out)
        seqs += s.total_sequences
        cls += s.classified
        looks += s.table_lookups
    assert torch.equal(torch.cat(parts), ref)
    assert (seqs, cls, looks) == (st.total_sequences, st.classified, st.table_lookups)
    # permute the fragments (offsets stay sorted: the reads are gathered into a new buffer)
    g = torch.Generator(device=big["dev"])
    g.manual_seed(5)
    perm = torch.randperm(N, generator=g, device=big["dev"])
    idx = (perm[:, None] * (2 * L) + torch.arange(2 * L, device=big["dev"])[None, :]).reshape(-1)
    pb = torch.cat([big["bases"][idx], big["bases"][-64:]]).contiguous()
    assert torch.equal(_classify(big, pb, offs, N, True), ref[perm])


def test_reverse_complement_of_every_read_gives_the_same_record(big):
    """Minimizers are canonical, runs of equal minimizers and the taxon tallies are symmetric under
    reversal: call, k-mer count, clade hits and hit groups of a read equal those of its reverse
    complement (single-end, so that no mate order is involved).  Reads with an N are left out:
    kraken2 calls a k-mer ambiguous when the N lies in its LAST l bases, which is not symmetric."""
    torch = big["torch"]
    offs = big["offs2"][: N + 1].contiguous()
    ref = _classify(big, big["bases"], offs, N, False)
    b = big["bases"][: N * L].reshape(N, L).flip(1)
    comp = torch.full((256,), 78, dtype=torch.uint8, device=big["dev"])
    for x, y in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[x] = y
    rc = torch.cat([comp[b.reshape(-1).long()], big["bases"][-64:]]).contiguous()
    got = _classify(big, rc, offs, N, False)
    clean = (b != 78).all(dim=1)
    assert 0.8 * N < int(clean.sum()) < N
    assert torch.equal(got[clean], ref[clean])


def test_long_reads_at_byte_offsets_beyond_4_gib(big):
    """configs[3] shape: 620 k reads of 10 kb = 6.2 GB of bases in ONE batch, so sequence offsets pass
    2^32.  The reads at the far end must classify exactly as they do from a compact copy at offset 0."""
    torch = big["torch"]
    dev = big["dev"]
    n, ln = 620_000, 10_000
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    bases = torch.empty(n * ln + 64, dtype=torch.uint8, device=dev)
    step = 20_000
    for i in range(0, n, step):  # filled piecewise: randint needs 8 bytes per element
        m = min(step, n - i)
        bases[i * ln:(i + m) * ln] = big["acgt"][torch.randint(0, 4, (m * ln,), generator=g, device=dev)]
    bases[n * ln:] = 65
    # some of the far reads carry sequence that is in the table (first reads of the shared fixture)
    src = big["bases"][: 150 * 1000]
    bases[(n - 500) * ln:(n - 500) * ln + src.numel()] = src
    offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * ln).contiguous()
    assert int(offs[-1]) > 2 ** 32
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    big["eng"].classify_device(bases.data_ptr(), offs.data_ptr(), n, False, 0.0, out.data_ptr(), long_reads=True)
    tail_n = 2_000
    tail = bases[(n - tail_n) * ln:].clone().contiguous()
    toffs = (torch.arange(tail_n + 1, dtype=torch.int64, device=dev) * ln).contiguous()
    tout = torch.zeros((tail_n, 4), dtype=torch.int32, device=dev)
    big["eng"].classify_device(tail.data_ptr(), toffs.data_ptr(), tail_n, False, 0.0, tout.data_ptr(), long_reads=True)
    torch.cuda.synchronize()
    assert torch.equal(out[n - tail_n:], tout)
    assert (out[:, 1] == ln - 35 + 1).all()
    assert int((tout[:, 0] != 0).sum()) >= 10  # the planted sequence was found


def test_ont_lognormal_batch_at_bench_size_equals_the_oracle(big, oracle_db, oracle_bases_per_second):
    """configs[3] at bench.py --ont shape: 400 k reads, length ~ lognormal(8.8, 0.85) clipped to
    [200, 200000] (N50 ~ 10 kb, ~3.7 Gbases in ONE launch, fragments handed out one by one), a slice of
    them carrying sequence that is in the table, 0.1 % N: every record equals the CPU oracle's."""
    from nohuman_amd.dist import usable_cpu_count
    torch = big["torch"]
    dev = big["dev"]
    n = 400_000
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    lens = torch.exp(torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 0.85 + 8.8)
    lens = lens.clamp(200, 200000).to(torch.int64)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1])
    assert 2.5e9 < total < 5e9
    bases = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    step = 200_000_000
    for i in range(0, total, step):
        m = min(step, total - i)
        bases[i:i + m] = big["acgt"][torch.randint(0, 4, (m,), generator=g, device=dev)]
        nm = torch.rand(m, generator=g, device=dev) < 0.001
        bases[i:i + m][nm] = 78
    bases[total:] = 65
    # every 40th read gets 3 kb of sequence whose minimizers are in the table, at a random place
    src = big["bases"][: 150 * 20_000]  # the fixture's inserted reads, back to back
    starts = offs[:-1][::40]
    room = (lens[::40] - 3000).clamp(min=0)
    at = starts + (torch.rand(starts.numel(), generator=g, device=dev, dtype=torch.float64) * room.double()).long()
    ok = lens[::40] >= 3000
    sel = torch.randint(0, src.numel() - 3000, (starts.numel(),), generator=g, device=dev)
    idx = torch.arange(3000, device=dev)[None, :]
    dst = (at[ok][:, None] + idx).reshape(-1)
    bases[dst] = src[(sel[ok][:, None] + idx).reshape(-1)]
    offs = offs.contiguous()
    out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    big["eng"].classify_device(bases.data_ptr(), offs.data_ptr(), n, False, 0.0, out.data_ptr(), cnt.data_ptr(),
                               long_reads=True)
    torch.cuda.synchronize()
    # the oracle takes the whole batch when this box does that within ORACLE_SECONDS, else the longest prefix
    offs_h = offs.cpu().numpy().astype(np.uint64)
    budget = int(oracle_bases_per_second * ORACLE_SECONDS)
    n_cmp = n if total <= budget else max(20_000, int(np.searchsorted(offs_h, budget)) - 1)
    print("oracle leg: %d of %d reads (%.2f of %.2f Gbases), oracle at %.0f Mbases/s on this box"
          % (n_cmp, n, int(offs_h[n_cmp]) / 1e9, total / 1e9, oracle_bases_per_second / 1e6))
    exp, lookups = oracle_db.classify(bases[:int(offs_h[n_cmp])].cpu().numpy(), offs_h[: n_cmp + 1], False, 0.0,
                                      threads=usable_cpu_count())
    rec = out[:n_cmp].cpu().numpy().view(np.uint32)
    for i, f in enumerate(("call", "total_kmers", "clade_hits", "hit_groups")):
        bad = np.nonzero(rec[:, i] != exp[f])[0]
        assert bad.size == 0, "%s differs at %s" % (f, bad[:5])
    c = cnt.tolist()
    assert c[0] == n and c[2] == total
    assert 0.01 * n < c[1] < 0.1 * n
    if n_cmp == n:
        assert c[3] == int(lookups.sum()) and c[1] == int((exp["call"] != 0).sum())
    else:  # the records of the rest at least carry the k-mer count their length implies
        rest = out[n_cmp:, 1].cpu().numpy().view(np.uint32)
        assert np.array_equal(rest, np.maximum(lens[n_cmp:].cpu().numpy() - 34, 0).astype(np.uint32))
    srt = torch.sort(lens, descending=True).values
    n50 = int(srt[torch.searchsorted(torch.cumsum(srt, 0), total // 2)])
    assert 8_000 < n50 < 20_000
