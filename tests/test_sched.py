"""The claim map of a launch (nh_device.h Sched; guided self-scheduling with a fixed map, NOHUMAN_SCHED=off = flat):
whatever the launch size, chunk size, mate count and grid, the claims tile [0, n_frag) exactly once, in
order, no chunk larger than c0, the sizes never growing.  Host logic only."""
import ctypes as C

import numpy as np
import pytest


def sched(n, c0, mates, waves):
    from nohuman_amd import _lib
    L = _lib.lib()
    out = (C.c_uint64 * 8)()
    L.nh_debug_sched.restype = None
    L.nh_debug_sched(C.c_uint64(n), C.c_uint32(c0), C.c_int(mates), C.c_uint64(waves), out)
    return dict(zip(("n0", "n01", "total", "base1", "base2", "c0", "c1", "c2"), list(out)))


def ranges(sc, n):
    """what claim_range() in nh_kernels.hip computes for every claim index: (begin, count) arrays"""
    i = np.arange(sc["total"], dtype=np.int64)
    c = np.where(i < sc["n0"], sc["c0"], np.where(i < sc["n01"], sc["c1"], sc["c2"])).astype(np.int64)
    beg = np.where(i < sc["n0"], i * sc["c0"],
                   np.where(i < sc["n01"], sc["base1"] + (i - sc["n0"]) * sc["c1"], sc["base2"] + (i - sc["n01"]) * sc["c2"]))
    return beg, np.minimum(c, n - beg)


@pytest.mark.parametrize("mates", [1, 2])
def test_claims_tile_the_launch(mates, monkeypatch):
    monkeypatch.delenv("NOHUMAN_SCHED", raising=False)
    rng = np.random.default_rng(7)
    sizes = [1, 2, 3, 5, 63, 64, 65, 1000, 4096, 99_999, 1_000_000, 2_500_000] + [int(x) for x in rng.integers(1, 300_000, 40)]
    for n in sizes:
        for c0 in ([1, 2, 4, 6, 24, 31] if mates == 2 else [1, 4, 8, 32, 63]):
            for waves in (4, 1024, 5120):
                sc = sched(n, c0, mates, waves)
                beg, cnt = ranges(sc, n)
                assert beg[0] == 0 and np.array_equal(beg[1:], np.cumsum(cnt)[:-1]), (n, c0, waves, sc)
                assert int(cnt.sum()) == n and cnt.min() >= 1 and cnt.max() <= sc["c0"]
                assert sc["c0"] >= sc["c1"] >= sc["c2"] >= 1
                # sizes never grow, and only the very last chunk of the launch may be short
                assert np.all(np.diff(cnt[:-1]) <= 0)
                assert set(np.unique(cnt[:-1]).tolist()) <= {sc["c0"], sc["c1"], sc["c2"]}


def test_tail_is_fine_grained_for_the_bench_shapes(monkeypatch):
    monkeypatch.delenv("NOHUMAN_SCHED", raising=False)
    for n, c0, mates in ((1_000_000, 32, 1), (2_500_000, 24, 2)):
        sc = sched(n, c0, mates, 5120)
        _, cnt = ranges(sc, n)
        assert sc["c1"] == c0 // 2 and sc["c2"] == (6 if mates == 2 else 12)
        tail = n - sc["base1"]
        assert 0.5 * 5120 * c0 * 0.7 <= tail <= 5120 * c0  # about three quarters of a chunk per wave
        assert cnt[-2] == sc["c2"]
    monkeypatch.setenv("NOHUMAN_SCHED", "off")  # the flat map
    sc = sched(1_000_000, 32, 1, 5120)
    assert sc["n0"] == sc["total"] - (1 if 1_000_000 % 32 else 0) or sc["c1"] == sc["c2"] == 32
    assert set(np.unique(ranges(sc, 1_000_000)[1]).tolist()) == {32}
