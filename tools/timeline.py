"""Per-wave timeline of k_classify_short (KArgs::timeline, 100 MHz timestamps): when waves start, get their
first claim, finish their first probe phase, take their last chunk and end.  usage: timeline.py [pe|se] [reads]
Needs an engine built with -DNH_TIMELINE (make -C nohuman_amd/csrc CXXFLAGS+=-DNH_TIMELINE): the product build ignores
NH_TIMELINE_PTR."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nohuman_amd import Engine
dev = torch.device("cuda", 0)
shape = sys.argv[1] if len(sys.argv) > 1 else "pe"
reads = int(sys.argv[2]) if len(sys.argv) > 2 else (5_000_000 if shape == "pe" else 1_000_000)
paired = shape == "pe"
mates = 2 if paired else 1
n = reads // mates
cap = 1_431_655_765
L = 150
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
eng = Engine.synthetic(cap, int(cap * 0.7), depth=30, seed=20250101)
g = torch.Generator(device=dev); g.manual_seed(11)
bases = acgt[torch.randint(0, 4, (reads * L + 64,), generator=g, device=dev)].contiguous()
offs = (torch.arange(reads + 1, dtype=torch.int64, device=dev) * L).contiguous()
res = torch.empty((n, 4), dtype=torch.int32, device=dev)
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
NW = 256 * 8 * 4
tl = torch.zeros((NW, 32), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step():
    eng.classify_device(bases.data_ptr(), offs.data_ptr(), n, paired, 0.0, res.data_ptr(), cnt.data_ptr(), st)
for _ in range(3): step()
torch.cuda.synchronize()
os.environ["NH_TIMELINE_PTR"] = hex(tl.data_ptr())
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); step(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
t = tl.cpu().numpy()
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("%s %d reads: launch %.3f ms (events), %d waves recorded, kernel span %.1f us" % (shape, reads, ms, len(t), us(t[:, 5].max())))
def row(name, v):
    q = np.percentile(v, [0, 1, 10, 50, 90, 99, 100])
    print("  %-34s min %8.1f  p1 %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f  p99 %8.1f  max %8.1f" % ((name,) + tuple(q)))
row("wave start (us after first)", us(t[:, 0]))
row("first claim returned", us(t[:, 1]))
row("  claim latency", (t[:, 1] - t[:, 0]) / 100.0)
row("first batch encoded", us(t[t[:, 2] != 0, 2]))
row("first probe phase done", us(t[t[:, 3] != 0, 3]))
row("last chunk taken", us(t[t[:, 4] != 0, 4]))
row("wave end", us(t[:, 5]))
row("chunks per wave", t[:, 6].astype(float))
end = us(t[:, 5])
span = end.max()
hist, edges = np.histogram(end, bins=20, range=(0, span))
print("  wave end histogram (20 bins over the span):", hist.tolist())
# waves still running over time: utilisation of the tail
for frac in (0.5, 0.8, 0.9, 0.95, 0.98, 0.99):
    print("  at %4.0f %% of the span (%7.1f us) %5d of %d waves still run" % (100 * frac, frac * span, int((end > frac * span).sum()), len(t)))
busy = (t[:, 5] - t[:, 0]).sum() / 100.0
print("  sum of wave lifetimes / (waves x span) = %.3f" % (busy / (len(t) * span)))
for x in range(8):
    m = t[:, 7] == x
    if m.any():
        print("  XCC %d: %4d waves, start p50 %6.1f, end p50 %7.1f max %7.1f, chunks mean %.2f" % (x, m.sum(), np.median(us(t[m, 0])), np.median(end[m]), end[m].max(), t[m, 6].mean()))

# per-chunk view: duration of a chunk = next chunk's start (or the wave's end) - its start, by chunk size
ch = t[:, 8:32]
durs = {}
for w in range(len(t)):
    k = int(min(t[w, 6], 24))
    for i in range(k):
        st_i, sz = ch[w, i] >> 8, int(ch[w, i] & 255)
        en = (ch[w, i + 1] >> 8) if i + 1 < k else t[w, 5]
        durs.setdefault((sz, i + 1 == k), []).append((en - st_i) / 100.0)
for (sz, last), v in sorted(durs.items()):
    v = np.array(v)
    print("  chunks of %2d fragments%s: %6d, duration p10 %6.1f p50 %6.1f p90 %6.1f us  (%.2f us per fragment)" % (
        sz, " (a wave's last, incl. drain)" if last else "                            ", len(v), *np.percentile(v, [10, 50, 90]), np.median(v) / sz))
# when does the work run out, and what is still being worked on then
last_start = (ch.max(axis=1) >> 8)
t_out = us(last_start.max())
print("  last chunk of the launch started at %.1f us; waves still running then: %d; span %.1f us" % (t_out, int((end > t_out).sum()), span))
