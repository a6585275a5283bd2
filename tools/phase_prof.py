"""Per-phase wave-cycle breakdown of k_classify (NH_PHASE_PROF=1 variant), bench workload."""
import os, sys
os.environ["NH_PHASE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nohuman_amd import Engine
dev = torch.device("cuda", 0)
cap = 1_431_655_765
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=20250101)
n, L = 1_000_000, 150
paired = "--se" not in sys.argv
mates = 2 if paired else 1
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
bases = acgt[torch.randint(0, 4, (n * mates * L + 64,), device=dev)].contiguous()
offs = (torch.arange(n * mates + 1, dtype=torch.int64, device=dev) * L).contiguous()
if "--hit" in sys.argv:  # every read "human": its minimizers are in the table (1 % of the bases mutated afterwards)
    eng.add_sequences(bases.data_ptr(), offs.data_ptr(), n * mates, 30)
    m = torch.rand(n * mates * L, device=dev) < 0.01
    sub = acgt[torch.randint(0, 4, (n * mates * L,), device=dev)]
    bases[: n * mates * L] = torch.where(m, sub, bases[: n * mates * L])
res = torch.empty((n, 4), dtype=torch.int32, device=dev)
cnt = torch.zeros(16, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(3):
    cnt.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.classify_device(bases.data_ptr(), offs.data_ptr(), n, paired, 0.0, res.data_ptr(), cnt.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
c = cnt.tolist()
ms = e0.elapsed_time(e1)
names = ["0 ->tile start", "1 base wait+encode", "2 lmer+window(LDS)", "3 runs+compact", "4 hash pass",
         "5 probe loop", "6 post tiles", "7 end of turn", "8 ->fragment top", "9 fragment header",
         "10 descriptor write", "11 probe rounds (count, not cycles)"]
tot = sum(c[4:])
print("kernel %.3f ms (instrumented), %d fragments, lookups %d" % (ms, c[0], c[3]))
for i, nm in enumerate(names):
    print("  %-22s %6.2f%%  %8.0f cycles/fragment" % (nm, 100.0 * c[4 + i] / tot, c[4 + i] / c[0]))
print("  total %.0f cycles/fragment/wave" % (tot / c[0]))

