"""Kernel time against launch size (fixed overhead a + per-read cost b): bench shapes, one process.
usage: size_scaling.py [pe|se] ...   env knobs (NOHUMAN_SCHED, ...) apply as set by the caller"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nohuman_amd import Engine
dev = torch.device("cuda", 0)
cap = 1_431_655_765
L = 150
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=20250101)
g = torch.Generator(device=dev); g.manual_seed(11)
NMAX = 10_000_000
pool = [acgt[torch.randint(0, 4, (NMAX * L + 64,), generator=g, device=dev)].contiguous() for _ in range(2)]
offs = (torch.arange(NMAX + 1, dtype=torch.int64, device=dev) * L).contiguous()
res = torch.empty((NMAX, 4), dtype=torch.int32, device=dev)
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for shape in (sys.argv[1:] or ["pe", "se"]):
    paired = shape == "pe"
    mates = 2 if paired else 1
    pts = []
    for reads in (125_000, 250_000, 500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000):
        n = reads // mates
        steps = max(5, min(40, 20_000_000 // reads))
        def step(i):
            eng.classify_device(pool[i % 2].data_ptr(), offs.data_ptr(), n, paired, 0.0, res.data_ptr(), cnt.data_ptr(), st)
        best = 1e9
        for rep in range(3):
            for i in range(2): step(i)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps): step(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / steps)
        pts.append((reads, best))
        print("%s %9d reads per launch: %8.4f ms  %7.1f Mreads/s" % (shape, reads, best, reads / best / 1e3), flush=True)
    # least squares a + b * reads over the four largest sizes
    import numpy as np
    x = np.array([p[0] for p in pts[-4:]], dtype=float); y = np.array([p[1] for p in pts[-4:]])
    b, a = np.polyfit(x, y, 1)
    print("%s fit over the four largest: %.4f ms + %.5f ms per 1000 reads (asymptote %.1f Mreads/s)" % (shape, a, b * 1e3, 1 / b / 1e3))
