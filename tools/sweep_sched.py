"""Sweeps launch-scheduling settings (NOHUMAN_SCHED / NOHUMAN_FRAG_CHUNK: read when an engine is opened, and again by
Engine.reload_launch_knobs(), which every setting here calls) on the
bench shapes in ONE process, interleaved passes (boxes and processes differ by a few percent; only numbers
of one call compare).  usage: sweep_sched.py [--passes 3] [--steps 20] [shape ...]   shapes: pe se hit ont"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nohuman_amd import Engine

dev = torch.device("cuda", 0)
argv = sys.argv[1:]
def opt(name, default):
    if name in argv:
        i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return type(default)(v)
    return default
passes, steps = opt("--passes", 3), opt("--steps", 20)
shapes = argv or ["pe", "se"]
cap = 1_431_655_765
L = 150
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)

def make(shape):
    eng = Engine.synthetic(cap, int(cap * 0.7) - (80_000_000 if shape == "hit" else 0), depth=30, seed=20250101)
    paired = shape in ("pe", "hit", "pechunk")
    mates = 2 if paired else 1
    n = {"pe": 2_500_000, "se": 1_000_000, "hit": 1_000_000, "ont": 200_000, "sechunk": 1_000_000, "pechunk": 2_500_000, "setail": 1_000_000}[shape]
    g = torch.Generator(device=dev); g.manual_seed(11)
    if shape == "ont":
        lens = torch.exp(torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 0.85 + 8.8).clamp(200, 200000).to(torch.int64)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev); offs[1:] = torch.cumsum(lens, 0)
    else:
        offs = torch.arange(n * mates + 1, dtype=torch.int64, device=dev) * L
    offs = offs.contiguous()
    total = int(offs[-1])
    pool = []
    for b in range(2):
        bases = acgt[torch.randint(0, 4, (total + 64,), generator=g, device=dev)].contiguous()
        if shape == "hit":
            nh = n // 2 * mates
            eng.add_sequences(bases.data_ptr(), offs.data_ptr(), nh, 30)
            he = nh * L
            m = torch.rand(he, generator=g, device=dev) < 0.01
            bases[:he] = torch.where(m, acgt[torch.randint(0, 4, (he,), generator=g, device=dev)], bases[:he])
        pool.append(bases)
    res = torch.empty((n, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    return dict(eng=eng, paired=paired, mates=mates, n=n, offs=offs, pool=pool, res=res, cnt=cnt, ont=shape == "ont")

def run(w, env):
    for k in ("NOHUMAN_SCHED", "NOHUMAN_FRAG_CHUNK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    torch.cuda.synchronize()
    w["eng"].reload_launch_knobs()
    st = torch.cuda.current_stream().cuda_stream
    def step(i):
        w["eng"].classify_device(w["pool"][i % 2].data_ptr(), w["offs"].data_ptr(), w["n"], w["paired"], 0.0,
                                 w["res"].data_ptr(), w["cnt"].data_ptr(), st, long_reads=w["ont"])
    for i in range(5): step(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): step(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

settings = {
    "pe": [("default 12,6,100,100", {}), ("off (flat)", {"NOHUMAN_SCHED": "off"}), ("12,4,50,50", {"NOHUMAN_SCHED": "12,4,50,50"}),
           ("12,4,150,150", {"NOHUMAN_SCHED": "12,4,150,150"}), ("12,2,100,100", {"NOHUMAN_SCHED": "12,2,100,100"}),
           ("12,6,100,100", {"NOHUMAN_SCHED": "12,6,100,100"}), ("8,4,100,100", {"NOHUMAN_SCHED": "8,4,100,100"}),
           ("16,8,100,100", {"NOHUMAN_SCHED": "16,8,100,100"})],
    "se": [("default 28,12,100,100", {}), ("off (flat)", {"NOHUMAN_SCHED": "off"}), ("28,12,50,50", {"NOHUMAN_SCHED": "28,12,50,50"}),
           ("28,12,150,150", {"NOHUMAN_SCHED": "28,12,150,150"}), ("28,12,200,200", {"NOHUMAN_SCHED": "28,12,200,200"}),
           ("28,12,100,200", {"NOHUMAN_SCHED": "28,12,100,200"}), ("28,12,200,100", {"NOHUMAN_SCHED": "28,12,200,100"}),
           ("28,4,100,100", {"NOHUMAN_SCHED": "28,4,100,100"}), ("28,8,100,100", {"NOHUMAN_SCHED": "28,8,100,100"}),
           ("28,20,100,100", {"NOHUMAN_SCHED": "28,20,100,100"}), ("16,8,100,100", {"NOHUMAN_SCHED": "16,8,100,100"}),
           ("40,12,100,100", {"NOHUMAN_SCHED": "40,12,100,100"}), ("20,4,150,150", {"NOHUMAN_SCHED": "20,4,150,150"})],
    "setail": [("default 28,12,100,100", {})] + [(x, {"NOHUMAN_SCHED": x}) for x in ("12,4,100,100", "12,4,60,100", "16,4,100,100", "20,4,100,100", "28,4,100,30",
                                                                                       "28,4,100,50", "28,8,100,50", "20,8,100,100", "24,8,100,100", "28,12,100,60")],
    "hit": [("default 12,6,100,100", {}), ("off (flat)", {"NOHUMAN_SCHED": "off"}), ("12,6,50,50", {"NOHUMAN_SCHED": "12,6,50,50"}),
            ("12,6,200,200", {"NOHUMAN_SCHED": "12,6,200,200"}), ("12,2,100,100", {"NOHUMAN_SCHED": "12,2,100,100"}), ("8,4,150,150", {"NOHUMAN_SCHED": "8,4,150,150"})],
    "ont": [("default", {})],
    "sechunk": [("default (60)", {}), ("40", {"NOHUMAN_FRAG_CHUNK": "40"}), ("48", {"NOHUMAN_FRAG_CHUNK": "48"}), ("56", {"NOHUMAN_FRAG_CHUNK": "56"}),
                ("60", {"NOHUMAN_FRAG_CHUNK": "60"}), ("28", {"NOHUMAN_FRAG_CHUNK": "28"}), ("24", {"NOHUMAN_FRAG_CHUNK": "24"}),
                ("20", {"NOHUMAN_FRAG_CHUNK": "20"}), ("16", {"NOHUMAN_FRAG_CHUNK": "16"}), ("12", {"NOHUMAN_FRAG_CHUNK": "12"})],
    "pechunk": [("default (24)", {}), ("28", {"NOHUMAN_FRAG_CHUNK": "28"}), ("20", {"NOHUMAN_FRAG_CHUNK": "20"}), ("16", {"NOHUMAN_FRAG_CHUNK": "16"}),
                ("12", {"NOHUMAN_FRAG_CHUNK": "12"}), ("30", {"NOHUMAN_FRAG_CHUNK": "30"})],
}
for shape in shapes:
    w = make(shape)
    reads = w["n"] * w["mates"]
    # the chip's first ~40 ms under load are a power-management transient (profiles/r05_launch_series.txt): launches are 10-30 %
    # slower there.  Every sweep starts behind it, and the passes keep the GPU busy back to back.
    t_wake = time.perf_counter()
    while time.perf_counter() - t_wake < 0.25:
        run(w, {})
    rows = {name: [] for name, _ in settings[shape]}
    for p in range(passes):
        for name, env in settings[shape]:
            rows[name].append(run(w, env))
    print("shape %s: %d fragments per launch, %d steps per measurement" % (shape, w["n"], steps))
    for name, _ in settings[shape]:
        ms = rows[name]
        print("  %-16s ms/launch %s   Mreads/s %s" % (name, " ".join("%.4f" % x for x in ms),
                                                     " ".join("%.1f" % (reads / x / 1e3) for x in ms)))
    sys.stdout.flush()
    w["eng"].close(); del w; torch.cuda.empty_cache()
