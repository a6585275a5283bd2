"""Launch-to-launch spread of the short classify launches: N launches of one shape back to back on one
stream, the GPU time of EVERY launch (HIP events between them), and the clocks sampled beside it (sysfs pp_dpm_sclk /
pp_dpm_mclk / current power state, every few milliseconds from a helper thread; `rocm-smi --showclocks` once before and
after).  Prints the series in blocks of ten, and the mean of the first 20, of launches 100-200 and of the last 100.
    python tools/launch_series.py se|pe|hit|wide|pe250 [launches=400] [idle_ms_before=0]"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nohuman_amd import Engine

shape = sys.argv[1] if len(sys.argv) > 1 else "se"
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 400
idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cap = 4_400_000_011 if shape == "wide" else 1_431_655_765
L = 250 if shape == "pe250" else 150
paired = shape != "se"
mates = 2 if paired else 1
n_frag = {"se": 1_000_000, "pe": 2_500_000, "hit": 1_000_000, "wide": 1_000_000, "pe250": 600_000}[shape]
hit = 0.5 if shape == "hit" else 0.0
n_keys = int(cap * 0.7) - (int(hit * n_frag * mates * 39.0 * 4) if hit else 0)
eng = Engine.synthetic(cap, n_keys, depth=30, seed=20250101, device=0)
acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
offsets = (torch.arange(n_frag * mates + 1, dtype=torch.int64, device=dev) * L).contiguous()
pool = []
for b in range(4):
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + b)
    bases = acgt[torch.randint(0, 4, (n_frag * mates * L + 64,), generator=g, device=dev)].contiguous()
    if hit:
        nh = int(hit * n_frag) * mates
        eng.add_sequences(bases.data_ptr(), offsets.data_ptr(), nh, 30)
    pool.append(bases)
results = torch.empty((n_frag, 4), dtype=torch.int32, device=dev)
counters = torch.zeros(4, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception as ex:
        return repr(ex)


def sysfs_clocks():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            try:
                txt = open(os.path.join(card, name)).read()
                cur = [ln for ln in txt.splitlines() if ln.strip().endswith("*")]
                out[os.path.basename(os.path.dirname(card)) + "." + name[7:]] = cur[0].split(":")[1].strip(" *") if cur else txt.strip()[:40]
            except OSError:
                pass
    return out


print("clocks before:", sysfs_clocks(), flush=True)
print(smi()[:1500], flush=True)
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), sysfs_clocks()))
        time.sleep(0.002)


torch.cuda.synchronize()
if idle_ms:
    time.sleep(idle_ms / 1e3)
th = threading.Thread(target=sampler)
th.start()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch + 1)]
t0 = time.perf_counter()
ev[0].record(stream)
for i in range(n_launch):
    eng.classify_device(pool[i % 4].data_ptr(), offsets.data_ptr(), n_frag, paired, 0.0, results.data_ptr(), counters.data_ptr(), stream.cuda_stream)
    ev[i + 1].record(stream)
torch.cuda.synchronize()
t1 = time.perf_counter()
stop = True
th.join()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n_launch)]
print("shape %s: %d launches of %d fragments in %.1f ms wall" % (shape, n_launch, n_frag, (t1 - t0) * 1e3))
for i in range(0, n_launch, 10):
    print("%4d: %s" % (i, " ".join("%.3f" % x for x in ms[i:i + 10])))
mean = lambda v: sum(v) / max(len(v), 1)
print("mean first 20 %.4f ms | launches 100-200 %.4f | last 100 %.4f | min %.4f | max after the first 50 %.4f" % (
    mean(ms[:20]), mean(ms[100:200]), mean(ms[-100:]), min(ms), max(ms[50:]) if n_launch > 50 else max(ms)))
seen = []
for t, c in samples:
    key = tuple(sorted(c.items()))
    if not seen or seen[-1][1] != key:
        seen.append((t - t0, key))
print("clock states seen while the launches ran (%d samples; seconds from the first launch):" % len(samples))
for t, key in seen[:40]:
    print("  %+.4f s  %s" % (t, dict(key)))
print("clocks after:", sysfs_clocks())
print(smi()[:1500])
eng.close()
