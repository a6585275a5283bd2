"""SURVEY §8b threading row: "nh_classify_batch is thread-safe per engine across distinct streams / batches"
(the reference's caller is single-threaded at the call, /root/reference/src/main.rs:270; a host that owns its I/O is not).

Round 6:
  * an engine has 16 launch slots (scheduling counters, deferral bitmap, long-read item buffers).  Rounds 1-5 took
    `launch_seq % 16` and documented "more than 16 in flight is not supported" with nothing to detect it.  A launch now
    waits -- on the device -- for the launch that used its slot before it: more than 16 in flight on any number of
    streams is ordered, never mixed.  The tests park every stream behind a spinning kernel so that ALL launches are in
    flight before the first one runs, on 3 and 5 streams (16 is a multiple of neither: launch i and i + 16 sit on
    different streams).
  * four host threads, a stream each, on ONE engine, at the same time.
Every launch's records equal the oracle's, and the caller's counters are the sums."""
import threading

import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu


def _batches(toy, toy_oracle, seed):
    """six kinds of launch: short single reads, pairs, reads of two and three tiles, long reads that are cut"""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(seed)
    pool = synth.sample_reads(rng, genomes, 1500, length=150, len_jitter=8)
    pool_pe = synth.sample_reads(rng, genomes, 800, length=150, paired=True, len_jitter=8)
    mid = synth.sample_reads(rng, genomes, 300, length=300, len_jitter=100)
    src = b"".join(genomes[k] for k in sorted(genomes))
    longs = []
    for _ in range(24):
        ln = int(rng.integers(3000, 9000))
        s = b"".join(src[int(st):int(st) + 700] for st in rng.integers(0, len(src) - 700, ln // 700 + 1))[:ln]
        longs.append(synth.mutate(rng, s, 0.01, 0.002, 0.0))
    out = []
    for kind, reads, paired, conf, long_reads in (
            ("se", [pool[i] for i in rng.integers(0, len(pool), 40_000)], False, 0.0, False),
            ("pe", [pool_pe[i] for i in rng.integers(0, len(pool_pe), 20_000)], True, 0.1, False),
            ("mid", [mid[i] for i in rng.integers(0, len(mid), 6_000)], False, 0.0, False),
            ("mixed", [pool[i] for i in rng.integers(0, len(pool), 9_000)] + [mid[i] for i in rng.integers(0, len(mid), 900)], False, 0.05, False),
            ("long", longs + [longs[i] for i in rng.integers(0, len(longs), 40)], False, 0.0, True),
            ("se_small", [pool[i] for i in rng.integers(0, len(pool), 700)], False, 0.5, False)):
        bases, offs = orc.pack_reads(reads, paired)
        exp, lookups = toy_oracle.classify(bases, offs, paired, conf)
        out.append(dict(kind=kind, bases=bases, offs=offs, paired=paired, conf=conf, long=long_reads, exp=exp,
                        n=len(reads), lookups=int(lookups.sum()), n_bases=int(offs[-1])))
    return out


def _to_device(torch, b):
    dev = torch.device("cuda:0")
    b["d_bases"] = torch.from_numpy(np.concatenate([b["bases"], np.full(64, 65, np.uint8)])).to(dev)
    b["d_offs"] = torch.from_numpy(b["offs"].astype(np.int64)).to(dev)


def _hold(torch, streams, ms=60):
    """park every stream behind a kernel that spins for about `ms`: what is queued meanwhile is all in flight at once"""
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(ms * 2.0e6))


def _check(torch, launches, cnt, what):
    torch.cuda.synchronize()
    tot = np.zeros(4, dtype=np.int64)
    for i, (b, out) in enumerate(launches):
        got = out.cpu().numpy().view(np.uint32)
        for j, f in enumerate(("call", "total_kmers", "clade_hits", "hit_groups")):
            bad = np.flatnonzero(got[:, j] != b["exp"][f])
            assert bad.size == 0, "%s: launch %d (%s): %d of %d records differ in %s, first at %d: got %r, expected %r; %d records never written" % (
                what, i, b["kind"], bad.size, b["n"], f, bad[0], got[bad[0]].tolist(), [int(b["exp"][g][bad[0]]) for g in
                                                                                         ("call", "total_kmers", "clade_hits", "hit_groups")],
                int((got == 0xFFFFFFFF).all(axis=1).sum()))
        tot += (b["n"], int((b["exp"]["call"] != 0).sum()), b["n_bases"], b["lookups"])
    assert cnt.cpu().numpy().tolist() == tot.tolist(), what


@pytest.fixture(scope="module")
def work(toy, toy_oracle):
    import torch
    bs = _batches(toy, toy_oracle, 606)
    for b in bs:
        _to_device(torch, b)
    return bs


@pytest.mark.parametrize("n_streams", [1, 3, 5])
def test_forty_launches_in_flight_on_several_streams(toy_engine, work, n_streams):
    import torch
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    # (the outputs are made and filled BEFORE the streams are parked: torch fills on its default stream, which is not
    #  ordered with the launches' streams -- a fill queued beside a parked launch can land after it)
    launches = []
    for i in range(40):
        b = work[(i * 5 + i // 7) % len(work)]
        launches.append((b, torch.full((b["n"], 4), -1, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    _hold(torch, streams)
    for i, (b, out) in enumerate(launches):
        toy_engine.classify_device(b["d_bases"].data_ptr(), b["d_offs"].data_ptr(), b["n"], b["paired"], b["conf"],
                                   out.data_ptr(), cnt.data_ptr(), streams[i % n_streams].cuda_stream, long_reads=b["long"])
    _check(torch, launches, cnt, "%d streams" % n_streams)


def test_four_host_threads_one_engine(toy_engine, work):
    """each thread: its own stream, its own outputs, 24 launches; all four start together behind a barrier while their
    streams are parked, so the threads' enqueues interleave and 96 launches are in flight"""
    import torch
    dev = torch.device("cuda:0")
    T, PER = 4, 24
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    plan = []
    for t in range(T):
        mine = []
        for i in range(PER):
            b = work[(t * 2 + i * 5 + i // 6) % len(work)]
            mine.append((b, torch.full((b["n"], 4), -1, dtype=torch.int32, device=dev)))
        plan.append(mine)
    torch.cuda.synchronize()
    _hold(torch, streams, ms=120)
    gate = threading.Barrier(T)
    errors = []

    def worker(t):
        try:
            gate.wait()
            for b, out in plan[t]:
                toy_engine.classify_device(b["d_bases"].data_ptr(), b["d_offs"].data_ptr(), b["n"], b["paired"], b["conf"],
                                           out.data_ptr(), cnt.data_ptr(), streams[t].cuda_stream, long_reads=b["long"])
        except Exception as e:  # noqa: BLE001 -- reported below, in the main thread
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    _check(torch, [l for mine in plan for l in mine], cnt, "four threads")


def test_threads_on_the_host_buffer_entry_serialise(toy, toy_oracle, toy_engine):
    """nh_classify_batch (host buffers, blocking) shares one staging area per engine: calls from several threads take
    the engine's mutex one after the other and each gets its own answer"""
    _, _, _, genomes, _ = toy
    rng = np.random.default_rng(11)
    jobs = []
    for t in range(4):
        reads = synth.sample_reads(rng, genomes, 600 + 50 * t, length=150, paired=bool(t & 1), len_jitter=20)
        bases, offs = orc.pack_reads(reads, bool(t & 1))
        exp, _ = toy_oracle.classify(bases, offs, bool(t & 1), 0.1)
        jobs.append((bases, offs, bool(t & 1), exp))
    got = [None] * 4

    def worker(t):
        for _ in range(5):
            got[t] = toy_engine.classify(jobs[t][0], jobs[t][1], jobs[t][2], 0.1)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for t in range(4):
        for f in ("call", "total_kmers", "clade_hits", "hit_groups"):
            assert np.array_equal(got[t][f], jobs[t][3][f]), (t, f)
