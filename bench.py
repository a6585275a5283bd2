#!/usr/bin/env python3
"""bench.py -- Mreads/s classified on 150 bp paired-end synthetic reads against an HPRC.r2-like
table resident in HBM (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path (one k_classify launch through the C ABI entry
nh_classify_batch_device) over one batch of synthetic read pairs already resident in HBM.
Reads shard across ranks with the database replicated; the only collective
is the final all-reduce of the classified counts (RCCL via torch.distributed "nccl").

Prints ONE JSON line on rank 0 (contract in the round brief), including
  "roofline":     algorithmic bytes (sum len + 64*D + 16 per fragment) per
                  launch / average kernel duration measured with HIP events on the launch stream,
  "cpu_baseline": the CPU oracle (oracle/k2_oracle.c, kind "port": kraken2 itself is not on the
                  box) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:35)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=2_500_000,
                    help="read pairs per step (= per launch) per GPU; the default 20 steps then cover the "
                         "50 M pairs of BASELINE.json configs[2]")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--single-end", action="store_true", help="config[1]: 150 bp single-end")
    ap.add_argument("--capacity", type=int, default=1_431_655_765,
                    help="hash table cells (HPRC.r2-like default: ~5.7 GB at load 0.7)")
    ap.add_argument("--load", type=float, default=0.7)
    ap.add_argument("--n-rate", type=float, default=0.0, help="per-base probability of 'N'")
    ap.add_argument("--pool", type=int, default=4, help="distinct batches cycled through")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--confidence", type=float, default=0.0)
    ap.add_argument("--hit-frac", type=float, default=0.0,
                    help="fraction of the fragments of every batch made 'human': their minimizers are "
                         "inserted into the table before the timed region, then 1%% of their bases mutated")
    ap.add_argument("--ont", action="store_true",
                    help="config[3] shape: single-end long reads, length ~ lognormal(8.8, 0.85) in "
                         "[200, 200000] (N50 ~ 10 kb); --pairs = number of reads")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: this process must use ONE HIP runtime (torch's bundled copy)
    import torch.distributed as dist

    import nohuman_amd
    from nohuman_amd import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    paired = not (args.single_end or args.ont)
    mates = 2 if paired else 1
    n_frag = args.pairs
    L = args.read_len

    # ---- database: synthetic HPRC.r2-like table built directly in this GPU's HBM -------------
    n_keys = int(args.capacity * args.load)
    if args.hit_frac > 0:  # leave room for the minimizers of the "human" reads: final load = --load
        n_keys = max(1, n_keys - int(args.hit_frac * args.pairs * (1 if (args.single_end or args.ont) else 2)
                                     * 39.0 * args.read_len / 150.0 * args.pool))
    t0 = time.time()
    eng = Engine.synthetic(args.capacity, n_keys, depth=30, seed=20250101, device=local_rank)
    t_db = time.time() - t0

    # ---- synthetic batches resident in HBM (iid uniform ACGT) -----------
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    n_seq = n_frag * mates
    if args.ont:
        g0 = torch.Generator(device=dev)
        g0.manual_seed(4)
        lens = torch.exp(torch.randn(n_seq, generator=g0, device=dev, dtype=torch.float64) * 0.85 + 8.8)
        lens = lens.clamp(200, 200000).to(torch.int64)
        offsets = torch.zeros(n_seq + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(lens, 0)
        offsets = offsets.contiguous()
    else:
        offsets = (torch.arange(n_seq + 1, dtype=torch.int64, device=dev) * L).contiguous()
    total_bases = int(offsets[-1].item())
    pool = []
    for b in range(args.pool):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 * (rank + 1) + b)
        idx = torch.randint(0, 4, (total_bases + 64,), generator=g, device=dev, dtype=torch.int64)
        bases = acgt[idx].contiguous()
        del idx
        if args.n_rate > 0:
            m = torch.rand(bases.shape, generator=g, device=dev) < args.n_rate
            bases[m] = 78
        if args.hit_frac > 0:
            n_hit_seq = int(args.hit_frac * n_frag) * mates
            if n_hit_seq:
                eng.add_sequences(bases.data_ptr(), offsets.data_ptr(), n_hit_seq, 30)
                hit_end = int(offsets[n_hit_seq].item())
                m = torch.rand(hit_end, generator=g, device=dev) < 0.01
                sub = acgt[torch.randint(0, 4, (hit_end,), generator=g, device=dev)]
                bases[:hit_end] = torch.where(m, sub, bases[:hit_end])
        pool.append(bases)
    info = eng.info  # after the inserts: size / capacity is the realised load factor
    results = torch.empty((n_frag, 4), dtype=torch.int32, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step(i):
        eng.classify_device(pool[i % len(pool)].data_ptr(), offsets.data_ptr(), n_frag, paired,
                            args.confidence, results.data_ptr(), counters.data_ptr(),
                            stream.cuda_stream, long_reads=args.ont)

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    counters.zero_()
    torch.cuda.synchronize()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)

    # ---- totals: the one collective of the path (classified-count all-reduce) ------------------
    from nohuman_amd.dist import reduce_counters
    (frags, classified, nbases, lookups), elapsed_max = reduce_counters(counters, elapsed)
    reads_total = frags * mates
    value = reads_total / elapsed_max / 1e6

    # per-launch algorithmic bytes on this rank
    c = [int(x) for x in counters.tolist()]
    alg_bytes_launch = (c[2] + 64 * c[3] + 16 * c[0]) / max(args.steps, 1)
    achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    out = {
        "metric": "Mreads/sec classified (HPRC.r2 DB, 150bp PE)" if not args.ont else "Mreads/sec classified (ONT)",
        "value": round(value, 3),
        "unit": "Mreads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / max(args.steps, 1) * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": (("ONT-like long reads (lognormal, %d bases per step), " % total_bases if args.ont else "")
                         + ("hit fraction %.2f, " % args.hit_frac if args.hit_frac else "") +
                         "%d x %d bp %s reads per step per GPU, iid uniform ACGT, resident in HBM; "
                         "synthetic HPRC.r2-like hash table (capacity %d cells = %.2f GB, load %.2f, "
                         "k=%d l=%d) replicated per GPU; confidence %g"
                         % (n_frag * mates, L, "paired-end" if paired else "single-end",
                            info.capacity, info.capacity * 4 / 1e9, info.size / info.capacity,
                            info.k, info.l, args.confidence)),
            "fragments_per_step": n_frag,
            "paired": paired,
            "parallelism": "reads sharded over %d GPU(s), DB replicated" % world,
            "classified_fraction": classified / max(frags, 1),
            "lookups_per_read": lookups / max(reads_total, 1),
            "db_build_seconds": round(t_db, 2),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(n_frag, paired, L, info.capacity, bool(args.ont or args.hit_frac or args.n_rate)),
            # informative: the limit this gather-bound kernel actually runs into is the fabric's random
            # request rate (profiles/traffic.json), not bytes: requests per launch / kernel time vs ceiling
            "fabric_request_frac": request_rate_frac(n_frag, paired, L, info.capacity, kernel_ms,
                                                     bool(args.ont or args.hit_frac or args.n_rate)),
            "kernel": "k_classify",
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": int(alg_bytes_launch),
        },
    }

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only) --------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(eng, pool[0], offsets, mates, paired, args, np, results, step)
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def measured_traffic(n_frag, paired, read_len, capacity, variant=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), if
    they were taken on exactly this workload; null otherwise.  bench.py cannot host the counter
    passes itself: the guide requires them in separate profiler runs."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    key = {"fragments_per_step": n_frag, "paired": paired, "read_len": read_len, "capacity": capacity}
    if variant:
        return None
    if t.get("workload") != key:
        return None
    return t.get("traffic_bytes_per_launch")


def request_rate_frac(n_frag, paired, read_len, capacity, kernel_ms, variant=False):
    """Fabric read requests per second of k_classify (PMC count per launch from profiles/traffic.json over
    the kernel time measured here) as a fraction of the rate a pure random gather sustains on the chip."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    key = {"fragments_per_step": n_frag, "paired": paired, "read_len": read_len, "capacity": capacity}
    if variant or t.get("workload") != key or not t.get("fabric_read_requests_per_launch") or kernel_ms <= 0:
        return None
    return round(t["fabric_read_requests_per_launch"] / (kernel_ms * 1e-3) / t["fabric_request_ceiling_per_s"], 4)


def cpu_baseline(eng, bases_dev, offsets_dev, mates, paired, args, np, results, step):
    """Times oracle/k2_oracle.c (pthreads, all host cores) on the first fragments of batch 0 with
    the very same table (downloaded from HBM), and checks the GPU results on that sample."""
    from oracle import oracle as orc
    from nohuman_amd.dist import usable_cpu_count
    cores = usable_cpu_count()
    cells = eng.download_table()
    info = eng.info
    odb = orc.OracleDB(eng.opts_image(), eng.taxonomy_image(), cells=cells,
                       header=(info.capacity, info.size, info.key_bits, info.value_bits))
    del cells
    chunk = 262144
    done = 0
    spent = 0.0
    n_frag = args.pairs
    host = None
    outs = []
    while done < n_frag and spent < args.cpu_seconds:
        n = min(chunk, n_frag - done)
        o = offsets_dev[done * mates:(done + n) * mates + 1].cpu().numpy().astype(np.uint64)
        host = bases_dev[int(o[0]):int(o[-1])].cpu().numpy()
        offs = o - o[0]
        t0 = time.perf_counter()
        exp, _ = odb.classify(host, offs, paired, args.confidence, threads=cores)
        spent += time.perf_counter() - t0
        outs.append(exp)
        done += n
        chunk = min(chunk * 2, 1 << 20)
    # parity of the GPU results on the sample (the oracle as checker)
    step(0)
    import torch
    torch.cuda.synchronize()
    got = results[:done].cpu().numpy().view(np.uint32)
    exp = np.concatenate(outs)
    ok = (np.array_equal(got[:, 0], exp["call"]) and np.array_equal(got[:, 1], exp["total_kmers"])
          and np.array_equal(got[:, 2], exp["clade_hits"]) and np.array_equal(got[:, 3], exp["hit_groups"]))
    return {
        "value": round(done * mates / spent / 1e6, 4),
        "unit": "Mreads/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d fragments (%d reads) of batch 0, same table; oracle/k2_oracle.c on %d "
                  "pthreads (kraken2 binary not on the box); GPU==oracle on sample: %s"
                  % (done, done * mates, cores, ok),
    }


if __name__ == "__main__":
    main()
