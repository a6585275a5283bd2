"""Same-process A/B of BUILDS of the library: every .so named on the command line is loaded beside the others (ctypes, one HIP
runtime), opens its own synthetic table, and classifies the SAME device-resident batches in interleaved passes at steady state
(boxes and processes differ by a few percent; only numbers of one call compare).
usage: ab_libs.py [--passes 4] [--steps 20] [--shapes pe,se,hit,ont] name=path.so [name=path.so ...]
  e.g.  ab_libs.py plain=nohuman_amd/libnohuman_engine.so nt=tools/ab_engine_nt.so        (make -C nohuman_amd/csrc ab-nt)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: one HIP runtime in the process)

from nohuman_amd import _lib  # noqa: E402

argv = sys.argv[1:]


def opt(name, default):
    if name in argv:
        i = argv.index(name)
        v = argv[i + 1]
        del argv[i:i + 2]
        return type(default)(v)
    return default


passes, steps = opt("--passes", 4), opt("--steps", 20)
shapes = opt("--shapes", "pe,se,hit,ont").split(",")
libs = [a.split("=", 1) for a in argv]
if not libs:
    raise SystemExit(__doc__)
_lib._preload_hip_runtime()
P = C.c_void_p


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.nh_open_synthetic.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(P)]
    L.nh_synthetic_add_sequences.argtypes = [P, P, P, C.c_uint64, C.c_uint32, P]
    L.nh_classify_batch_device.argtypes = [P, P, P, C.c_uint64, C.c_uint32, C.c_double, P, P, P, P, P]
    L.nh_close.argtypes = [P]
    L.nh_last_error.restype = C.c_char_p
    return L


dev = torch.device("cuda", 0)
cap, RL = 1_431_655_765, 150
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
LIBS = [(name, load(path)) for name, path in libs]


def chk(L, rc):
    if rc != 0:
        raise RuntimeError(L.nh_last_error().decode())


def make(shape):
    paired = shape in ("pe", "hit")
    mates = 2 if paired else 1
    n = {"pe": 2_500_000, "se": 1_000_000, "hit": 1_000_000, "ont": 200_000, "pe250": 600_000}[shape]
    rl = 250 if shape == "pe250" else RL
    if shape == "pe250":
        paired, mates = True, 2
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    if shape == "ont":
        lens = torch.exp(torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 0.85 + 8.8).clamp(200, 200000).to(torch.int64)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        offs[1:] = torch.cumsum(lens, 0)
    else:
        offs = torch.arange(n * mates + 1, dtype=torch.int64, device=dev) * rl
    offs = offs.contiguous()
    total = int(offs[-1])
    pool = [acgt[torch.randint(0, 4, (total + 64,), generator=g, device=dev)].contiguous() for _ in range(2)]
    engines = []
    for name, L in LIBS:
        h = P()
        chk(L, L.nh_open_synthetic(cap, int(cap * 0.7) - (80_000_000 if shape == "hit" else 0), 30, 20250101, 0, C.byref(h)))
        if shape == "hit":
            for b in pool:
                chk(L, L.nh_synthetic_add_sequences(h, b.data_ptr(), offs.data_ptr(), n // 2 * mates, 30, None))
        engines.append(h)
    if shape == "hit":  # 1 % substitutions in the "human" half, as bench.py does
        he = n // 2 * mates * rl
        for b in pool:
            m = torch.rand(he, generator=g, device=dev) < 0.01
            b[:he] = torch.where(m, acgt[torch.randint(0, 4, (he,), generator=g, device=dev)], b[:he])
    res = torch.empty((n, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    return dict(n=n, mates=mates, paired=paired, offs=offs, pool=pool, engines=engines, res=res, cnt=cnt, ont=shape == "ont")


def run(w, k, n_steps):
    name, L = LIBS[k]
    st = torch.cuda.current_stream().cuda_stream
    flags = (1 if w["paired"] else 0) | (2 if w["ont"] else 0)

    def step(i):
        chk(L, L.nh_classify_batch_device(w["engines"][k], w["pool"][i % 2].data_ptr(), w["offs"].data_ptr(), w["n"], flags, 0.0,
                                          w["res"].data_ptr(), None, None, w["cnt"].data_ptr(), st))
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_steps):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_steps


for shape in shapes:
    w = make(shape)
    reads = w["n"] * w["mates"]
    t_wake = time.perf_counter()
    while time.perf_counter() - t_wake < 0.25:  # behind the chip's power-management transient (profiles/r05_launch_series.txt)
        run(w, 0, 4)
    rows = {name: [] for name, _ in LIBS}
    sums = {}
    for p in range(passes):
        order = range(len(LIBS)) if p % 2 == 0 else reversed(range(len(LIBS)))
        for k in order:
            w["cnt"].zero_()
            rows[LIBS[k][0]].append(run(w, k, steps))
            sums[LIBS[k][0]] = tuple(w["cnt"].tolist()) + (int(w["res"][:, 0].to(torch.int64).sum()),)
    same = len(set(sums.values())) == 1
    print("shape %s: %d fragments per launch, %d steps per measurement; counters and call sums %s" % (
        shape, w["n"], steps, "identical across the builds" if same else "DIFFER: %r" % sums))
    base = sorted(rows[LIBS[0][0]])[len(rows[LIBS[0][0]]) // 2]
    for name, _ in LIBS:
        ms = rows[name]
        med = sorted(ms)[len(ms) // 2]
        print("  %-12s ms/launch %s   median %.4f = %.1f Mreads/s (%+.2f %% vs %s)" % (
            name, " ".join("%.4f" % x for x in ms), med, reads / med / 1e3, 100.0 * (base / med - 1.0), LIBS[0][0]))
    sys.stdout.flush()
    for (name, L), h in zip(LIBS, w["engines"]):
        L.nh_close(h)
    del w
    torch.cuda.empty_cache()
