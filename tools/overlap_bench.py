"""Two launches in flight on two streams vs one (hides the ramp-up and drain of a launch).
   gpurun -- python3 tools/overlap_bench.py [pairs_per_launch]"""
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/nohuman_amd') else os.getcwd())
import torch
from nohuman_amd import Engine


def main():
    dev = torch.device("cuda", 0)
    cap = 1_431_655_765
    eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=20250101)
    n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 150
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    pool = []
    for b in range(4):
        bases = acgt[torch.randint(0, 4, (n * 2 * L + 64,), device=dev)].contiguous()
        pool.append(bases)
    offs = (torch.arange(n * 2 + 1, dtype=torch.int64, device=dev) * L).contiguous()
    res = [torch.empty((n, 4), dtype=torch.int32, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def run(nstreams, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            s = streams[i % nstreams]
            eng.classify_device(pool[i % 4].data_ptr(), offs.data_ptr(), n, True, 0.0, res[i % 4].data_ptr(), 0, s.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    for ns in (1, 2, 1, 2):
        run(ns, 5)
        ms = run(ns, 40)
        print("%d stream(s): %.4f ms per step, %.1f Mreads/s" % (ns, ms, 2 * n / ms / 1e3))


if __name__ == "__main__":
    main()
