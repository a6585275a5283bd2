import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from nohuman_amd import Engine
dev = torch.device("cuda", 0)
cap = 1_431_655_765
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=20250101)
n, L = 1_000_000, 150
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
bases = acgt[torch.randint(0, 4, (n * 2 * L + 64,), device=dev)].contiguous()
offs = (torch.arange(n * 2 + 1, dtype=torch.int64, device=dev) * L).contiguous()
res = torch.empty((n, 4), dtype=torch.int32, device=dev)
per = 2 * (L - 35 + 1) + 1
toff = (torch.arange(n + 1, dtype=torch.int64, device=dev) * per).contiguous()
taxa = torch.empty(n * per + 1, dtype=torch.int32, device=dev)
def run(with_taxa, steps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        eng.classify_device(bases.data_ptr(), offs.data_ptr(), n, True, 0.0, res.data_ptr(), 0, 0,
                            taxa.data_ptr() if with_taxa else 0, toff.data_ptr() if with_taxa else 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for w in (False, True, False, True):
    run(w, 3); ms = run(w)
    print("kmer_taxa %s: %.3f ms per 1M pairs = %.1f Mreads/s" % (w, ms, 2 * n / ms / 1e3))
