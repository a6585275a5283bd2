#!/usr/bin/env python3
"""bench.py -- Mreads/s classified on 150 bp paired-end synthetic reads against an HPRC.r2-like
table resident in HBM (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path (one k_classify launch through the C ABI entry
nh_classify_batch_device) over one batch of synthetic read pairs already resident in HBM.
Reads shard across ranks with the database replicated; the only collective
is the final all-reduce of the classified counts (RCCL via torch.distributed "nccl").

`python bench.py --gpus N` works both ways: under torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) this process IS one rank; started directly with N > 1 it only spawns
the N rank processes (before anything here touches the GPU), relays rank 0's line and exits with
their status.

Prints ONE JSON line on rank 0 (contract in the round brief).  Everything the driver keeps is inside the objects it
preserves whole:
  "config":       the workload, and config.e2e (N=1 only): nh_run() files-in -> files-out on gzip pairs (configs[2] shape
                  at a stated scale) and on one gzip file of ONT-like reads (configs[3] shape, scaled) -- Mreads/s and wall
                  seconds only, never mixed into `value`;
  "roofline":     algorithmic bytes (sum len + 64*D + 16 per fragment) per launch / average kernel
                  duration measured with HIP events on the launch stream; value_two_streams: the same launches issued on two
                  alternating streams (what nh_run's stream slots do); roofline.variants (N=1 only, reduced step counts):
                  single-end configs[1], hit path, ONT-like long reads, a table of > 2^32 cells, 2 x 250 bp pairs -- frac,
                  kernel_ms, both values, checked against the oracle on a sample;
  "cpu_baseline": kraken2 itself with the reference's argv (src/main.rs:215-267) when a `kraken2` binary is on PATH
                  (kind "kraken2"), else the CPU oracle (oracle/k2_oracle.c, kind "port") timed on the host cores on a
                  bounded sample of the same workload.
The long descriptions (workload strings, per-stage traces, what each leg did) go to stderr and to bench_details.json
(directory NOHUMAN_BENCH_LOGDIR, default the current one), not into the line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import datetime

RDZV_TIMEOUT = datetime.timedelta(seconds=120)  # init_process_group / collectives: fail, do not hang
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:35)


def parse_args(argv=None):
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
    ap.add_argument("--wake-ms", type=float, default=80.0,
                    help="untimed launches of the same step before the warm-up steps until this much wall time has passed: the "
                         "chip's power-management transient after idle (profiles/r05_launch_series.txt); 0 = none")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the hit-path / single-end legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the nh_run files-in -> files-out leg")
    ap.add_argument("--e2e-pairs", type=int, default=5_000_000, help="pairs per gzip member of the e2e inputs")
    ap.add_argument("--e2e-reps", type=int, default=10, help="members per e2e input file")
    ap.add_argument("--e2e-distinct", type=int, default=5,
                    help="distinct members generated (they repeat in rotation: compressing 36 GB of text at level 6 takes "
                         "the host two minutes, and the default run has to stay within a few)")
    ap.add_argument("--e2e-multi-pairs", type=int, default=20_000_000,
                    help="--gpus N > 1: pairs PER DEVICE of the one-process nh_run(n_devices = N) leg (configs[4] shape, cyclic members)")
    ap.add_argument("--multi-child", default=None, help=argparse.SUPPRESS)  # internal: the fresh process of that leg
    ap.add_argument("--confidence", type=float, default=0.0)
    ap.add_argument("--hit-frac", type=float, default=0.0,
                    help="fraction of the fragments of every batch made 'human': their minimizers are "
                         "inserted into the table before the timed region, then 1%% of their bases mutated")
    ap.add_argument("--ont", action="store_true",
                    help="config[3] shape: single-end long reads, length ~ lognormal(8.8, 0.85) in "
                         "[200, 200000] (N50 ~ 10 kb); --pairs = number of reads")
    return ap.parse_args(argv)


# ---- direct start with --gpus N > 1: spawn the ranks, touch nothing else ---------------------------
def launch_ranks(args):
    """One child process per GPU with the torch.distributed environment set.  This parent never
    imports torch or calls HIP (a process that has initialised the GPU must not start ranks that
    replace it).  It WATCHES all ranks: the first rank that exits non-zero is named, the
    others are terminated (fresh children of this process -- nothing is re-executed) and the launcher exits
    non-zero within seconds instead of leaving rank 0 in its rendezvous until torch's timeout.  Every rank's
    stdout + stderr also go to bench_rank<r>.log (directory NOHUMAN_BENCH_LOGDIR, default: the current one);
    rank 0's JSON line is relayed to this process's stdout."""
    import signal
    import socket
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = os.environ.get("NOHUMAN_BENCH_LOGDIR", ".")
    os.makedirs(logdir, exist_ok=True)
    procs, logs, threads = [], [], []
    out0 = []

    def pump(r, stream, is_out):  # tee a rank's stream: log file, and stderr of the launcher / rank 0's stdout buffer
        for raw in iter(stream.readline, b""):
            logs[r].write(raw)
            logs[r].flush()
            if is_out and r == 0:
                out0.append(raw)
            elif not is_out:
                sys.stderr.buffer.write(raw)
                sys.stderr.buffer.flush()

    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        logs.append(open(os.path.join(logdir, "bench_rank%d.log" % r), "wb"))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        procs.append(p)
        for stream, is_out in ((p.stdout, True), (p.stderr, False)):
            t = threading.Thread(target=pump, args=(r, stream, is_out), daemon=True)
            t.start()
            threads.append(t)
    codes = [None] * args.gpus
    first_bad = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] and first_bad is None:
                    first_bad = r
        if first_bad is not None:
            break
        time.sleep(0.05)
    if first_bad is not None:
        sys.stderr.write("bench.py: rank %d of %d exited with status %d (see %s); stopping the other ranks\n"
                         % (first_bad, args.gpus, codes[first_bad], os.path.join(logdir, "bench_rank%d.log" % first_bad)))
        for r, p in enumerate(procs):  # exact process groups this launcher started, never a pattern
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        deadline = time.time() + 10
        for r, p in enumerate(procs):
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
            if codes[r] is None:
                codes[r] = p.returncode
    for t in threads:
        t.join(timeout=5)
    for f in logs:
        f.close()
    sys.stdout.write(b"".join(out0).decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c]
    for r, c in bad:
        if r != first_bad:
            sys.stderr.write("bench.py: rank %d of %d ended with status %d\n" % (r, args.gpus, c))
    return 1 if bad else 0


class Ctx:
    """Per-process state shared by the legs: torch, numpy, device, rank layout."""


def measure(cx, args, *, steps, warmup, single_end=False, ont=False, hit_frac=0.0, n_rate=0.0, pairs=None,
            keep=False, capacity=None, read_len=None, contract_first=False):
    """Builds the synthetic table and batches in this GPU's HBM and times `steps` launches.  Returns a
    dict with the numbers of one bench line; with keep=True also the live objects (engine, batches)."""
    torch, dist, np = cx.torch, cx.dist, cx.np
    from nohuman_amd import Engine
    dev = cx.dev
    paired = not (single_end or ont)
    mates = 2 if paired else 1
    n_frag = pairs if pairs is not None else args.pairs
    L = read_len or args.read_len

    # ---- database: synthetic HPRC.r2-like table built directly in this GPU's HBM -------------
    capacity = capacity or args.capacity
    n_keys = int(capacity * args.load)
    if hit_frac > 0:  # leave room for the minimizers of the "human" reads: final load = --load
        need = int(hit_frac * n_frag * mates * 39.0 * L / 150.0 * args.pool)
        if need > 0.9 * n_keys:  # (a toy --capacity: the inserts would fill the table and probe for ever)
            raise ValueError("a table of %d cells cannot take the %d minimizers of the hit-path variant" % (capacity, need))
        n_keys = max(1, n_keys - need)
    t0 = time.time()
    pin_db = os.environ.get("NOHUMAN_PIN_DB") if capacity == args.capacity and not hit_frac else None
    if pin_db:
        # a REAL database directory (hash.k2d / opts.k2d / taxo.k2d, e.g. HPRC.r2 of /root/reference/config.toml:1-7):
        # loaded as it is and used instead of the synthetic table -- its header is reported in config.database
        eng = Engine.open(pin_db, device=cx.dev_index)
    else:
        eng = Engine.synthetic(capacity, n_keys, depth=30, seed=20250101, device=cx.dev_index)
    t_db = time.time() - t0

    # ---- synthetic batches resident in HBM (iid uniform ACGT) -----------
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    n_seq = n_frag * mates
    if ont:
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
        g.manual_seed(1000 * (cx.rank + 1) + b)
        idx = torch.randint(0, 4, (total_bases + 64,), generator=g, device=dev, dtype=torch.int64)
        bases = acgt[idx].contiguous()
        del idx
        if n_rate > 0:
            m = torch.rand(bases.shape, generator=g, device=dev) < n_rate
            bases[m] = 78
        if hit_frac > 0:
            n_hit_seq = int(hit_frac * n_frag) * mates
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
                            stream.cuda_stream, long_reads=ont)

    def barrier():
        if cx.world > 1:
            dist.barrier()

    # The chip's first ~40 ms under load are a power-management transient: launches run 10-30 % slower there whatever their
    # size (profiles/r05_launch_series.txt: 1 M single reads 1.34 -> 1.027 ms, 2.5 M pairs 5.96 -> 4.755 ms, sclk within 3 %
    # all the while; afterwards launch-to-launch spread is +-1 %).  A run of nh_run keeps the GPU busy for seconds, so the
    # steady state is what a launch costs: the same launches are issued untimed until --wake-ms of GPU work have run, THEN
    # the W warm-up steps and the K timed steps follow as the contract says.
    # the contract's own sequence -- W warm-up steps, K timed, nothing before them -- is measured FIRST and reported
    # beside `value` (roofline.value_without_wake / kernel_ms_without_wake), so that rounds 1-4's numbers stay comparable.
    no_wake = None
    if contract_first and args.wake_ms > 0:
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(steps):
            step(warmup + i)
        e1.record(stream)
        torch.cuda.synchronize()
        no_wake = e0.elapsed_time(e1) / max(steps, 1)
    wake_launches = 0
    if args.wake_ms > 0:
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < args.wake_ms:
            step(wake_launches)
            wake_launches += 1
            if wake_launches % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    counters.zero_()
    torch.cuda.synchronize()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for i in range(steps):
        step(warmup + i)
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = ev0.elapsed_time(ev1) / max(steps, 1)

    # ---- the same launches on TWO alternating streams: what nh_run's stream slots do in production (a launch's tail
    # overlaps the next one's ramp).  Reported beside the single-stream value, never instead of it.
    s2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    res2 = [results, torch.empty_like(results)]
    counters2 = torch.zeros(4, dtype=torch.int64, device=dev)

    def step2(i):
        eng.classify_device(pool[i % len(pool)].data_ptr(), offsets.data_ptr(), n_frag, paired, args.confidence,
                            res2[i % 2].data_ptr(), counters2.data_ptr(), s2[i % 2].cuda_stream, long_reads=ont)

    # (the few hundred microseconds of synchronising and allocating above are enough of an idle moment to bring part of the
    #  transient back -- the profiler's per-dispatch times show it, profiles/r05_launch_series.txt --: the same wake phase here)
    if args.wake_ms > 0:
        tw = time.perf_counter()
        k = 0
        while (time.perf_counter() - tw) * 1e3 < args.wake_ms / 2:
            step2(k)
            k += 1
            if k % 4 == 0:
                torch.cuda.synchronize()
    for i in range(min(warmup, 2)):
        step2(i)
    torch.cuda.synchronize()
    barrier()
    t2 = time.perf_counter()
    for i in range(steps):
        step2(warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed2 = time.perf_counter() - t2
    del res2, counters2

    # ---- totals: the one collective of the path (classified-count all-reduce) ------------------
    from nohuman_amd.dist import gather_floats, reduce_counters
    (frags, classified, nbases, lookups), elapsed_max = reduce_counters(counters, elapsed)
    rank_kernel_ms = gather_floats(kernel_ms, dev)
    elapsed2_max = max(gather_floats(elapsed2, dev))
    reads_total = frags * mates
    value = reads_total / elapsed_max / 1e6
    value2 = reads_total / elapsed2_max / 1e6

    # per-launch algorithmic bytes on this rank
    c = [int(x) for x in counters.tolist()]
    alg_bytes_launch = (c[2] + 64 * c[3] + 16 * c[0]) / max(steps, 1)
    achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # which committed PMC record (profiles/traffic.json) belongs to this shape, if any
    wide = capacity >= (1 << 32)
    tkey = None
    if not n_rate and L == 250 and paired and not (ont or wide or hit_frac) and capacity == 1_431_655_765 and abs(args.load - 0.7) < 1e-9:
        tkey = "pe250"
    if n_rate == 0.001 and L == 150 and paired and not (ont or wide or hit_frac) and capacity == 1_431_655_765 and abs(args.load - 0.7) < 1e-9:
        tkey = "n"
    if not n_rate and L == 150 and abs(args.load - 0.7) < 1e-9:
        if ont:
            tkey = "ont"
        elif wide:
            tkey = "wide" if paired and not hit_frac else None
        elif capacity == 1_431_655_765:
            tkey = "hit" if (hit_frac == 0.5 and paired) else None if hit_frac else ("pe" if paired else "se")
    m = {
        "value": round(value, 3),
        "ms_per_step": round(elapsed_max / max(steps, 1) * 1e3, 4),
        "workload": ((("%d ONT-like single-end long reads per step per GPU (length lognormal(8.8, 0.85) in [200, 200000], "
                       "%d bases per step, mean %d)" % (n_frag, total_bases, total_bases // max(n_frag, 1))) if ont else
                      (("hit fraction %.2f, " % hit_frac if hit_frac else "") +
                       "%d x %d bp %s reads per step per GPU" % (n_frag * mates, L, "paired-end" if paired else "single-end")))
                     + ", iid uniform ACGT, resident in HBM; %s hash table (capacity %d cells = %.2f GB, "
                       "load %.2f, k=%d l=%d) replicated per GPU; confidence %g"
                     % ("REAL database %s:" % pin_db if pin_db else "synthetic HPRC.r2-like",
                        info.capacity, info.capacity * 4 / 1e9, info.size / info.capacity, info.k, info.l, args.confidence)),
        "database": {"source": pin_db or "synthetic (nh_open_synthetic: random keys at the stated load, 30-node chain taxonomy)",
                     "capacity": info.capacity, "size": info.size, "key_bits": info.key_bits, "value_bits": info.value_bits,
                     "node_count": info.node_count, "k": info.k, "l": info.l,
                     "spaced_seed_mask": "0x%x" % info.spaced_seed_mask, "toggle_mask": "0x%x" % info.toggle_mask,
                     "minimum_acceptable_hash_value": info.minimum_acceptable_hash_value,
                     "revcom_version": info.revcom_version},
        "fragments_per_step": n_frag,
        "paired": paired,
        "classified_fraction": classified / max(frags, 1),
        "lookups_per_read": lookups / max(reads_total, 1),
        "db_build_seconds": round(t_db, 2),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # PMC counters cannot be collected inside this run (the guide wants them in separate
            # rocprofv3 passes): these two come from the committed passes on exactly this workload
            "traffic": static_traffic(tkey, n_frag, "traffic_bytes_per_launch"),
            "traffic_source": "profiles/traffic.json workloads.%s (static: rocprofv3 --pmc passes of this workload, "
                              "not measured in this run)" % tkey,
            # true: the record was taken on other kernel sources than this run's (sha256 of nh_kernels.hip + nh_device.h)
            "traffic_stale": traffic_stale(tkey, n_frag),
            "fabric_request_frac": request_rate_frac(tkey, n_frag, kernel_ms),
            # the same static traffic over THIS run's kernel time, against the 6.29 TB/s the chip reaches on a copy
            "hbm_achievable_frac": hbm_frac(tkey, n_frag, kernel_ms),
            "kernel": "k_classify_short" if not (ont or L > 158) else "k_classify",
            "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_per_rank": [round(x, 4) for x in rank_kernel_ms],
            "wake_ms": args.wake_ms, "wake_launches": wake_launches,  # untimed launches before the W warm-up steps (see measure())
            # the contract's sequence with NO wake phase in front (W warm-ups, K timed; measured first): this rank's kernel time
            # and the Mreads/s it gives
            "kernel_ms_without_wake": round(no_wake, 4) if no_wake else None,
            "value_without_wake": round(n_frag * mates * cx.world / (no_wake * 1e-3) / 1e6, 3) if no_wake else None,
            "algorithmic_bytes_per_launch": int(alg_bytes_launch),
            # the same launches on two alternating streams (nh_run's slots): whole-job Mreads/s and its roofline fraction
            "value_two_streams": round(value2, 3),
            "frac_two_streams": round(alg_bytes_launch * steps / elapsed2_max / 1e9 / HBM_PEAK_GBS, 4) if elapsed2_max > 0 else None,
        },
    }
    live = dict(eng=eng, pool=pool, offsets=offsets, results=results, step=step, mates=mates, paired=paired,
                n_frag=n_frag, ont=ont)
    if keep:
        return m, live
    eng.close()
    return m, None


def main():
    args = parse_args()
    if args.multi_child:
        raise SystemExit(e2e_multi_child(args.multi_child))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    import numpy as np
    import torch  # first: this process must use ONE HIP runtime (torch's bundled copy)
    import torch.distributed as dist

    import nohuman_amd  # noqa: F401

    cx = Ctx()
    cx.torch, cx.dist, cx.np = torch, dist, np
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and cx.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, cx.world))
    if os.environ.get("NOHUMAN_BENCH_DRYRUN"):
        # CPU check of the launch plumbing (tests/test_dist.py): rendezvous over gloo, one all-reduce
        if os.environ.get("NOHUMAN_BENCH_FAIL_RANK") == str(cx.rank):  # test hook: this rank dies before the rendezvous
            raise SystemExit(7)
        if cx.world > 1:
            dist.init_process_group("gloo", timeout=RDZV_TIMEOUT)
        one = torch.ones(1, dtype=torch.int64)
        if cx.world > 1:
            dist.all_reduce(one)
        if cx.rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": cx.world, "ranks_counted": int(one.item()),
                              "world_size_seen": dist.get_world_size() if cx.world > 1 else 1}), flush=True)
        if cx.world > 1:
            dist.destroy_process_group()
        return
    # NOHUMAN_BENCH_ONE_GPU=1 (test mode for boxes with one GPU): every rank uses device 0 and the counters are
    # reduced over gloo -- RCCL refuses two ranks on one device.  Everything else of the N > 1 path is the real thing
    # (rank processes, sharding, barriers, max-over-ranks timing, the reductions, the rank table).
    cx.one_gpu = bool(os.environ.get("NOHUMAN_BENCH_ONE_GPU"))
    cx.dev_index = 0 if cx.one_gpu else cx.local_rank
    if cx.dev_index >= torch.cuda.device_count():
        raise SystemExit("rank %d: no GPU %d on this node (%d visible)" % (cx.rank, cx.dev_index,
                                                                           torch.cuda.device_count()))
    torch.cuda.set_device(cx.dev_index)
    cx.dev = torch.device("cuda", cx.dev_index)
    backend = None
    if os.environ.get("NOHUMAN_BENCH_FAIL_RANK") == str(cx.rank):  # test hook: this rank dies before the rendezvous
        raise SystemExit(7)
    if cx.world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank that never arrives is a diagnosis after two minutes, not torch's default half hour
        if cx.one_gpu:
            dist.init_process_group("gloo", timeout=RDZV_TIMEOUT)
        else:
            dist.init_process_group("nccl", device_id=cx.dev, timeout=RDZV_TIMEOUT)
        backend = dist.get_backend()

    m, live = measure(cx, args, steps=args.steps, warmup=args.warmup, single_end=args.single_end, ont=args.ont,
                      hit_frac=args.hit_frac, n_rate=args.n_rate, keep=True, contract_first=True)
    out = {
        "metric": "Mreads/sec classified (HPRC.r2 DB, 150bp PE)" if not args.ont else "Mreads/sec classified (ONT)",
        "value": m["value"],
        "unit": "Mreads/s",
        "n_gpus": cx.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": m["workload"],
            "fragments_per_step": m["fragments_per_step"],
            "paired": m["paired"],
            "parallelism": "reads sharded over %d GPU(s), DB replicated" % cx.world,
            "world_size_seen": dist.get_world_size() if cx.world > 1 else 1,
            "collective_backend": ("gloo (TEST MODE NOHUMAN_BENCH_ONE_GPU: all %d ranks share GPU 0)" % cx.world) if cx.one_gpu and backend
                                  else ("%s (RCCL %s)" % (backend, ".".join(map(str, torch.cuda.nccl.version()))))
                                  if backend else "none (single process)",
            "ranks": rank_table(cx, m["roofline"]["kernel_ms"]),
            "classified_fraction": m["classified_fraction"],
            "lookups_per_read": m["lookups_per_read"],
            "db_build_seconds": m["db_build_seconds"],
            "database": m["database"],
        },
        "roofline": m["roofline"],
    }
    solo = cx.rank == 0 and cx.world == 1
    details = {"headline_workload": m["workload"]}
    out["config"]["workload"] = short_workload(m)
    # ---- CPU baseline: kraken2 itself when the box has it, else the oracle on the host cores; bounded sample (rank 0, N=1 only)
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cx, args, live, args.cpu_seconds)
    # ---- e2e: files in -> files out through nh_run on the same engine (N=1 only) --------------------
    plain_pe = not (args.ont or args.single_end)
    if solo and not args.no_e2e and plain_pe:
        try:
            out["config"]["e2e"], details["e2e"] = e2e_leg(cx, args, live["eng"])
        except Exception as ex:  # the e2e leg must never cost the headline line
            out["config"]["e2e"] = {"error": repr(ex)[:300]}
        try:
            ont_nums, details["e2e_ont"] = e2e_ont_leg(cx, args, live["eng"])
            out["config"].setdefault("e2e", {}).update(ont_nums)
        except Exception as ex:
            out["config"].setdefault("e2e", {})["ont_error"] = repr(ex)[:300]
    # ---- --gpus N > 1: the PRODUCT's own multi-device path -- ONE process, nh_run(n_devices = N): lanes of the gzip reader
    # over the devices, peer copies, ncclCommInitAll, one writer per file -- which N single-device rank processes never
    # touch.  Rank 0 writes the inputs and the database directory now and runs the leg in a FRESH child
    # process once every rank has closed its engine and left the process group (below).
    multi = cx.rank == 0 and cx.world > 1 and not args.no_e2e and not (args.ont or args.single_end)
    if not multi:  # (rank 0 keeps its engine until the leg's database directory is written, below)
        live["eng"].close()
        del live
        torch.cuda.empty_cache()
    # ---- the output stage on its own: the GPU gzip encoder on FASTQ text (N=1 only) ------------------------------
    if solo and not args.no_e2e and plain_pe:
        try:
            gzl = gzip_leg(cx, args)
            details["gzip_encoder"] = gzl
            out["config"].setdefault("e2e", {})["gzip_encoder"] = {
                "kernel_GBps": gzl["kernel_GBps"], "wall_GBps": gzl["wall_GBps"], "ratio": gzl["ratio"],
                "inflates_to_the_text": gzl["inflates_to_the_text"], "zlib6_one_core_MBps": gzl["zlib6_one_core"]["MBps"],
                "zlib6_ratio": gzl["zlib6_one_core"]["ratio"]}
        except Exception as ex:
            out["config"].setdefault("e2e", {})["gzip_encoder"] = {"error": repr(ex)[:300]}
    # ---- variants at reduced step counts (N=1 only): inside `roofline`, numbers only ------------------------
    if solo and not args.no_variants and not (args.ont or args.single_end or args.hit_frac):
        variants = {}
        details["variants"] = {}
        for name, kw in (("se", dict(single_end=True, pairs=1_000_000)),            # configs[1]
                         ("hit", dict(hit_frac=0.5, pairs=1_000_000)),              # half of the fragments "human"
                         ("ont", dict(ont=True, pairs=200_000)),                    # configs[3] shape, 200 k reads a launch
                         # a table of more than 2^32 cells (64-bit cell positions in the kernels): 17.6 GB, four copies
                         ("wide", dict(pairs=1_000_000, capacity=4_400_000_011)),
                         # 2 x 250 bp Illumina pairs: longer than one tile (158 bases), so every chunk is left to the generic
                         # kernel k_classify -- the shape MiSeq / NovaSeq SP 2 x 250 runs meet
                         ("pe250", dict(pairs=600_000, read_len=250)),
                         # SURVEY 8d's "+N": every base replaced by N with p = 0.001 -- the ambiguity path (exact ambiguity
                         # flags, k-mers that are not looked up) on the headline shape
                         ("n", dict(pairs=1_000_000, n_rate=0.001))):
            try:
                vm, vlive = measure(cx, args, steps=20, warmup=3, keep=True, **kw)
            except ValueError as ex:  # (a variant this --capacity cannot hold: said, not run)
                variants[name] = {"skipped": str(ex)}
                continue
            chk = cpu_baseline(cx, args, vlive, 1.5, oracle_only=True)
            vlive["eng"].close()
            del vlive
            torch.cuda.empty_cache()
            r = vm["roofline"]
            variants[name] = {
                "frac": r["frac"], "kernel_ms": r["kernel_ms"], "value": vm["value"], "value_two_streams": r["value_two_streams"],
                "frac_two_streams": r["frac_two_streams"], "hbm_achievable_frac": r["hbm_achievable_frac"],
                "traffic": r["traffic"], "traffic_stale": r["traffic_stale"],
                "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "fragments_per_step": vm["fragments_per_step"],
                "classified_fraction": round(vm["classified_fraction"], 4), "lookups_per_read": round(vm["lookups_per_read"], 3),
                "gpu_equals_oracle": chk["gpu_equals_oracle"], "oracle_sample_fragments": chk["fragments"],
            }
            details["variants"][name] = {"workload": vm["workload"], "steps": 20, "warmup": 3, "wake_ms": args.wake_ms, "kernel": r["kernel"]}
        out["roofline"]["variants"] = variants
        # the driver's record keeps the SCALAR keys of `config` and `roofline` only: every number of the
        # variants' table again as roofline.<variant>_<what>
        for name, v in variants.items():
            for key, short in (("frac", "frac"), ("kernel_ms", "kernel_ms"), ("value", "value"), ("gpu_equals_oracle", "equals_oracle"),
                               ("hbm_achievable_frac", "achievable_frac"), ("frac_two_streams", "frac_two_streams")):
                if key in v:
                    out["roofline"]["%s_%s" % (name, short)] = v[key]
            if "skipped" in v:
                out["roofline"]["%s_skipped" % name] = v["skipped"][:120]
    if cx.world > 1:
        dist.barrier()  # (every other rank's engine is closed; they leave now, rank 0 goes on alone)
        dist.destroy_process_group()
    if multi:
        multi_spec = None
        try:
            multi_spec = e2e_multi_prepare(cx, args, live["eng"])
            live["eng"].close()
            del live
            torch.cuda.empty_cache()
            out["config"].setdefault("e2e", {})["multi"], details["e2e_multi"] = e2e_multi_run(multi_spec)
        except Exception as ex:
            out["config"].setdefault("e2e", {})["multi"] = {"error": repr(ex)[:300]}
        finally:
            if multi_spec:
                import shutil
                shutil.rmtree(multi_spec["tmp"], ignore_errors=True)
    flatten_e2e(out["config"])
    if cx.rank == 0:
        write_details(details)
    if cx.rank == 0:
        print(json.dumps(out), flush=True)


def flatten_e2e(config):
    """config.e2e is a nested object and the driver's record keeps scalars only: the numbers of DESIGN 6.3 again as
    config.e2e_<leg> (Mreads/s unless the name says otherwise)."""
    e = config.get("e2e")
    if not isinstance(e, dict):
        return
    def num(*path):
        v = e
        for k in path:
            if not isinstance(v, dict) or k not in v:
                return None
            v = v[k]
        return v
    flat = {
        "e2e_gzip_to_plain": num("gzip_to_plain", "value"), "e2e_gzip_to_plain_wall_s": num("gzip_to_plain", "wall_s"),
        "e2e_gzip_to_gzip": num("gzip_to_gzip", "value"), "e2e_gzip_to_gzip_wall_s": num("gzip_to_gzip", "wall_s"),
        "e2e_input_side": num("input_side_only", "value"), "e2e_input_side_wall_s": num("input_side_only", "wall_s"),
        "e2e_ont_gzip_to_gzip": num("ont_gzip_to_gzip", "value"), "e2e_ont_gzip_to_gzip_gbases": num("ont_gzip_to_gzip", "Gbases_per_s"),
        "e2e_ont_input_side": num("ont_input_side_only", "value"),
        "e2e_pairs": num("pairs"), "e2e_ont_reads": num("ont_gzip_to_gzip", "reads"), "e2e_host_threads": num("host_threads"),
        "e2e_reader": num("gzip_to_gzip", "reader") or num("gzip_to_plain", "reader"),
        "e2e_outputs_equal_inputs": (num("outputs_equal_inputs") is True and num("ont_gzip_to_gzip", "outputs_equal_inputs") is True)
        if num("outputs_equal_inputs") is not None else None,
        "e2e_host_reader_gzip_to_plain": num("readers_by_name", "host", "gzip_to_plain"),
        "e2e_host_reader_gzip_to_gzip": num("readers_by_name", "host", "gzip_to_gzip"),
        "e2e_host_reader_input_side": num("readers_by_name", "host", "input_side_only"),
        "e2e_gzip_encoder_kernel_GBps": num("gzip_encoder", "kernel_GBps"), "e2e_gzip_encoder_ratio": num("gzip_encoder", "ratio"),
        "e2e_error": num("error") or num("ont_error"),
    }
    for k, v in flat.items():
        if v is not None:
            config[k] = v
    m = e.get("multi")
    if isinstance(m, dict):
        for k, v in m.items():
            if not isinstance(v, (dict, list)):
                config["e2e_multi_%s" % k] = v


def short_workload(m):
    """config.workload: one short phrase (the full description goes to bench_details.json)"""
    return "%d x %s per step per GPU, iid ACGT resident in HBM; hash table %.2f GB (%s), load %.2f" % (
        m["fragments_per_step"], "read pairs" if m["paired"] else "single-end reads", m["database"]["capacity"] * 4 / 1e9,
        "REAL database" if not str(m["database"]["source"]).startswith("synthetic") else "synthetic HPRC.r2-like",
        m["database"]["size"] / max(m["database"]["capacity"], 1))


def write_details(details):
    """What the line no longer carries: stderr (one line) and bench_details.json"""
    try:
        txt = json.dumps(details)
        sys.stderr.write("bench-details: " + txt + "\n")
        sys.stderr.flush()
        logdir = os.environ.get("NOHUMAN_BENCH_LOGDIR", ".")
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "bench_details.json"), "w") as fo:
            fo.write(txt + "\n")
    except OSError:
        pass


def rank_table(cx, kernel_ms):
    """One entry per rank, in rank order: device name, kernel time of the rank's launches, host."""
    import socket
    torch, dist = cx.torch, cx.dist
    mine = {"rank": cx.rank, "device": torch.cuda.get_device_name(cx.dev), "kernel_ms": kernel_ms,
            "host": socket.gethostname()}
    if cx.world == 1:
        return [mine]
    rows = [None] * cx.world
    dist.all_gather_object(rows, mine)
    return rows


def _traffic_record(key, n_frag):
    """The committed rocprofv3 PMC record of a workload (profiles/traffic.json, written by scripts/profile.sh +
    make_profile_summary.py), if it was taken on exactly this shape."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, None
    w = t.get("workloads", {}).get(key) if key else None
    if not w or w.get("workload", {}).get("fragments_per_step") != n_frag:
        return None, t
    return w, t


def kernel_source_sha16():
    """sha256 of the classify kernels' sources, as scripts/make_profile_summary.py records it next to the PMC numbers"""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in ("nohuman_amd/csrc/nh_kernels.hip", "nohuman_amd/csrc/nh_device.h"):
            h.update(open(os.path.join(ROOT, f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def traffic_stale(key, n_frag):
    """True when the committed PMC record was taken on OTHER kernel sources than the ones this run was built from
    (static traffic from a superseded binary), None when there is no record."""
    w, _ = _traffic_record(key, n_frag)
    if not w:
        return None
    return w.get("kernel_source_sha16") != kernel_source_sha16()


def static_traffic(key, n_frag, field):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), if they were taken
    on exactly this workload; null otherwise.  bench.py cannot host the counter passes itself: the guide
    requires them in separate profiler runs."""
    w, _ = _traffic_record(key, n_frag)
    return w.get(field) if w else None


def request_rate_frac(key, n_frag, kernel_ms):
    """Fabric read requests per second of the classify kernel (PMC count per launch from profiles/traffic.json over
    the kernel time measured here) as a fraction of the rate a pure random gather sustains on the chip."""
    w, t = _traffic_record(key, n_frag)
    if not w or not w.get("fabric_read_requests_per_launch") or kernel_ms <= 0:
        return None
    return round(w["fabric_read_requests_per_launch"] / (kernel_ms * 1e-3) / t["fabric_request_ceiling_per_s"], 4)


def hbm_frac(key, n_frag, kernel_ms):
    w, t = _traffic_record(key, n_frag)
    if not w or not w.get("traffic_bytes_per_launch") or kernel_ms <= 0:
        return None
    return round(w["traffic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9 / t.get("hbm_achievable_gbs", 6290.0), 4)


def write_k2_db(db_dir, opts_image, taxo_image, header, cells):
    """A kraken2 database directory from the engine's images (nh_opts_image, nh_taxonomy_image, nh_table_download): the
    three files of /root/reference/src/lib.rs:120, in the layout of SURVEY.md A.1."""
    import struct
    os.makedirs(db_dir, exist_ok=True)
    with open(os.path.join(db_dir, "opts.k2d"), "wb") as fo:
        fo.write(opts_image)
    with open(os.path.join(db_dir, "taxo.k2d"), "wb") as fo:
        fo.write(taxo_image)
    with open(os.path.join(db_dir, "hash.k2d"), "wb") as fo:
        fo.write(struct.pack("<4Q", *[int(x) for x in header]))
        cells.tofile(fo)


def kraken2_argv(threads, db_dir, out_pattern, inputs, confidence=0.0, kraken_output="/dev/null"):
    """The argv nohuman builds for its subprocess (/root/reference/src/main.rs:210-267), token for token: --threads T --db DB
    --output /dev/null --confidence C [--paired] --unclassified-out <tmp>/kraken_out[#].fq inputs.  (Rust prints 0.0f32 as "0".)"""
    conf = ("%g" % confidence)
    argv = ["--threads", str(int(threads)), "--db", db_dir, "--output", kraken_output, "--confidence", conf]
    if len(inputs) == 2:
        argv.append("--paired")
    argv += ["--unclassified-out", out_pattern]
    return argv + list(inputs)


def time_kraken2(exe, threads, db_dir, inputs, tmp, confidence=0.0):
    """Runs the stock binary with the reference's argv; wall seconds, and what its own stderr summary says (the three
    integers of /root/reference/src/lib.rs:61-97 and the `processed in Ts` timer, which excludes the database load)."""
    import re
    out = os.path.join(tmp, "kraken_out#.fq" if len(inputs) == 2 else "kraken_out.fq")
    argv = [exe] + kraken2_argv(threads, db_dir, out, inputs, confidence)
    t = time.perf_counter()
    p = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    wall = time.perf_counter() - t
    for q in (out.replace("#", "_1"), out.replace("#", "_2"), out):
        if os.path.exists(q):
            os.remove(q)
    if p.returncode != 0:
        raise RuntimeError("kraken2 failed with stderr %s" % p.stderr[-500:])
    err = p.stderr.replace(",", "")
    m = re.search(r"(\d+) sequences \(([0-9.]+) Mbp\) processed in ([0-9.]+)s", err)
    c = re.search(r"(\d+) sequences classified", err)
    u = re.search(r"(\d+) sequences unclassified", err)
    return {"argv": argv, "wall_s": wall, "sequences": int(m.group(1)) if m else None, "own_seconds": float(m.group(3)) if m else None,
            "classified": int(c.group(1)) if c else None, "unclassified": int(u.group(1)) if u else None}


def sample_fastq(cx, live, n_frag, paths):
    """The first n_frag fragments of batch 0 as FASTQ (ids syn.<idx>[/1|/2], qualities 'I': SURVEY.md section 8d)"""
    torch = cx.torch
    mates = live["mates"]
    offs = live["offsets"][:n_frag * mates + 1].cpu().numpy()
    bases = live["pool"][0][:int(offs[-1])].cpu().numpy()
    for mate, path in enumerate(paths):
        with open(path, "wb") as fo:
            rows = []
            for i in range(n_frag):
                a, b = int(offs[i * mates + mate]), int(offs[i * mates + mate + 1])
                rows.append(b"@syn.%d%s\n" % (i, (b"/%d" % (mate + 1)) if mates == 2 else b""))
                rows.append(bases[a:b].tobytes())
                rows.append(b"\n+\n" + b"I" * (b - a) + b"\n")
                if len(rows) >= 300000:
                    fo.write(b"".join(rows))
                    rows = []
            fo.write(b"".join(rows))


def kraken2_baseline(cx, args, live, exe, cores):
    """SURVEY.md section 8d case 1: the stock kraken2 binary on the headline's first batch with exactly the argv nohuman builds,
    at `cores` threads and at 1 thread, against the SAME table (written to a tmpfs database directory from the engine's images).
    The same directory is what scripts/parity_vs_kraken2.sh takes at bench size."""
    import shutil
    import tempfile
    np, torch = cx.np, cx.torch
    eng, mates = live["eng"], live["mates"]
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="nh_bench_k2_", dir=base)
    try:
        info = eng.info
        db = os.path.join(tmp, "db")
        write_k2_db(db, eng.opts_image(), eng.taxonomy_image(), (info.capacity, info.size, info.key_bits, info.value_bits), eng.download_table())
        n_all = min(live["n_frag"], int(os.environ.get("NOHUMAN_BENCH_K2_FRAGMENTS", "2500000")))
        n_one = max(1, n_all // 10)
        paths = [os.path.join(tmp, "r_%d.fq" % (m + 1)) for m in range(mates)]
        sample_fastq(cx, live, n_all, paths)
        full = time_kraken2(exe, cores, db, paths, tmp, args.confidence)
        one_paths = [p + ".one" for p in paths]
        for p, q in zip(paths, one_paths):  # the first tenth for the one-thread figure
            with open(p, "rb") as fi, open(q, "wb") as fo:
                lines = 4 * n_one
                for k, ln in enumerate(fi):
                    if k >= lines:
                        break
                    fo.write(ln)
        one = time_kraken2(exe, 1, db, one_paths, tmp, args.confidence)
        # the GPU on the same sample: classified counts must agree (the per-read diff is scripts/parity_vs_kraken2.sh's job)
        live["step"](0)
        torch.cuda.synchronize()
        gpu_classified = int((live["results"][:n_all, 0] != 0).sum().item())
        keep = os.environ.get("NOHUMAN_BENCH_K2_KEEP")  # a directory: the database and the sample stay for the parity script
        if keep:
            shutil.copytree(tmp, keep, dirs_exist_ok=True)
        secs = full["own_seconds"] or full["wall_s"]
        return {
            "value": round(n_all * mates / secs / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "kraken2",
            "wall_value": round(n_all * mates / full["wall_s"] / 1e6, 4),
            "one_thread": {"value": round(n_one * mates / (one["own_seconds"] or one["wall_s"]) / 1e6, 4), "fragments": n_one},
            "fragments": n_all,
            "classified": full["classified"], "gpu_classified": gpu_classified,
            "classified_equal_gpu": full["classified"] == gpu_classified,
            "argv": " ".join(full["argv"]),
            "sample": "first %d fragments (%d reads) of batch 0 as FASTQ in tmpfs, the engine's own table written as a kraken2 database; "
                      "stock kraken2 with nohuman's argv (src/main.rs:215-267) at %d threads; value from its own 'processed in' timer "
                      "(database load excluded), wall_value from the wall clock" % (n_all, n_all * mates, cores),
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(cx, args, live, budget_s, oracle_only=False):
    """Times oracle/k2_oracle.c (pthreads, all host cores) on the first fragments of batch 0 with
    the very same table (downloaded from HBM), and checks the GPU results on that sample.  When a `kraken2` binary is on
    PATH (pin day) the stock binary is the baseline (kind "kraken2") and the port stays beside it."""
    import shutil
    np, torch = cx.np, cx.torch
    from oracle import oracle as orc
    from nohuman_amd.dist import usable_cpu_count
    eng, mates, paired = live["eng"], live["mates"], live["paired"]
    cores = usable_cpu_count()
    cells = eng.download_table()
    info = eng.info
    odb = orc.OracleDB(eng.opts_image(), eng.taxonomy_image(), cells=cells,
                       header=(info.capacity, info.size, info.key_bits, info.value_bits))
    del cells
    chunk = 65536 if budget_s < 5 else 262144
    if live.get("ont"):
        chunk = 8192  # ~80 Mbases per oracle call
    done = 0
    spent = 0.0
    n_frag = live["n_frag"]
    outs = []
    while done < n_frag and spent < budget_s:
        n = min(chunk, n_frag - done)
        o = live["offsets"][done * mates:(done + n) * mates + 1].cpu().numpy().astype(np.uint64)
        host = live["pool"][0][int(o[0]):int(o[-1])].cpu().numpy()
        offs = o - o[0]
        t0 = time.perf_counter()
        exp, _ = odb.classify(host, offs, paired, args.confidence, threads=cores)
        spent += time.perf_counter() - t0
        outs.append(exp)
        done += n
        chunk = min(chunk * 2, 1 << 20)
    # parity of the GPU results on the sample (the oracle as checker)
    live["step"](0)
    torch.cuda.synchronize()
    got = live["results"][:done].cpu().numpy().view(np.uint32)
    exp = np.concatenate(outs)
    ok = (np.array_equal(got[:, 0], exp["call"]) and np.array_equal(got[:, 1], exp["total_kmers"])
          and np.array_equal(got[:, 2], exp["clade_hits"]) and np.array_equal(got[:, 3], exp["hit_groups"]))
    del odb
    exe = None if oracle_only else shutil.which("kraken2")
    port = {
        "value": round(done * mates / spent / 1e6, 4),
        "unit": "Mreads/s",
        "cores": cores,
        "kind": "port",
        "gpu_equals_oracle": bool(ok),
        "fragments": done,
        "sample": "first %d fragments (%d reads) of batch 0, same table; oracle/k2_oracle.c on %d "
                  "pthreads (%s); GPU==oracle on sample: %s"
                  % (done, done * mates, cores, "kraken2 timed beside it" if exe else "kraken2 binary not on the box", ok),
    }
    if not exe:
        return port
    try:
        k2 = kraken2_baseline(cx, args, live, exe, cores)
    except Exception as ex:  # a binary that does not run must not cost the line: the port stands, the failure is named
        port["kraken2_error"] = repr(ex)[:300]
        return port
    k2["gpu_equals_oracle"] = bool(ok)
    k2["port"] = {k: port[k] for k in ("value", "cores", "fragments", "gpu_equals_oracle")}
    return k2


def _hash_file_ranges(path, ranges):
    """xxh3-64 of byte ranges of a file (read in 64 MB pieces; the hash releases the GIL)."""
    import xxhash
    out = []
    with open(path, "rb", buffering=0) as f:
        for off, ln in ranges:
            h = xxhash.xxh3_64()
            f.seek(off)
            left = ln
            while left > 0:
                b = f.read(min(left, 64 << 20))
                if not b:
                    break
                h.update(b)
                left -= len(b)
            out.append(h.intdigest() if left == 0 else None)
    return out


def gzip_leg(cx, args):
    """nh_gzip_gpu_file (nohuman_amd/csrc/nh_deflate.hip) on FASTQ text like the e2e leg's, host buffer in, gzip
    file out: kernel time by HIP events inside the library, wall time here, zlib -6 on one core beside it (what
    the reference's gzp runs per block, compression.rs:214-233).  The file is inflated by the library's own
    reader and compared with the text by digest."""
    import ctypes
    import shutil
    import tempfile
    import zlib
    import xxhash
    from nohuman_amd import _lib
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="nh_bench_gz_", dir=base)
    try:
        parts = []
        for m in range(3):
            p = os.path.join(tmp, "m%d.fq" % m)
            e2e_member(cx, 400_000, args.read_len, 1, m, p)
            parts.append(open(p, "rb").read())
            os.remove(p)
        unit = b"".join(parts)
        data = unit * 5
        L = _lib.lib()
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
        gz_path = os.path.join(tmp, "out.gz")
        best = None
        for _ in range(3):
            st = (ctypes.c_uint64 * 2)()
            t = time.perf_counter()
            if L.nh_gzip_gpu_file(cx.dev.index or 0, buf, len(data), os.fsencode(gz_path), st) != 0:
                raise RuntimeError(L.nh_last_error().decode(errors="replace"))
            dt = time.perf_counter() - t
            if best is None or st[1] < best[1]:
                best = (dt, int(st[1]), int(st[0]))
        dt, kernel_us, out_bytes = best
        plain = os.path.join(tmp, "back.fq")
        st3 = (ctypes.c_uint64 * 3)()
        rc = L.nh_gunzip_file(os.fsencode(gz_path), os.fsencode(plain), 8, 0, st3)
        same = rc == 0 and os.path.getsize(plain) == len(data) and \
            _hash_file_ranges(plain, [(0, len(data))])[0] == xxhash.xxh3_64(data).intdigest()
        sample = unit[:48 << 20]
        t = time.perf_counter()
        z = zlib.compress(sample, 6)
        dz = time.perf_counter() - t
        return {
            "workload": "%.2f GB of FASTQ text (150 bp reads, Illumina-style ids, binned qualities; %.0f MB distinct, in rotation), "
                        "host buffer -> one gzip member in tmpfs" % (len(data) / 1e9, len(unit) / 1e6),
            "kernel_GBps": round(len(data) / (kernel_us / 1e6) / 1e9, 2),
            "kernel_ms": round(kernel_us / 1e3, 2),
            "wall_GBps": round(len(data) / dt / 1e9, 2),
            "ratio": round(len(data) / out_bytes, 3),
            "inflates_to_the_text": bool(same),
            "zlib6_one_core": {"MBps": round(len(sample) / dz / 1e6, 1), "ratio": round(len(sample) / len(z), 3)},
            "what": "one wave per 64 KiB of text, dynamic Huffman blocks of 32 KiB, %s-way hash buckets in LDS (NOHUMAN_GZIP_WAYS); kernel time = HIP "
                    "events around the encoder's kernels of every 128 MiB chunk; wall includes staging, CRC-32 and the file"
                    % os.environ.get("NOHUMAN_GZIP_WAYS", "4"),
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_member(cx, n, L, tag, member, path):
    """One gzip member of a synthetic Illumina-like FASTQ file, written as plain text to `path`: per-read
    distinct ids in the instrument's style (lane / tile / x / y fields that change the way a sorted run's do),
    iid bases, and NovaSeq-style binned qualities (F : , #) that start to degrade at a per-read position --
    text that gzip -6 takes to about a quarter, like real reads, not the 6:1 of constant qualities."""
    torch = cx.torch
    dev = cx.dev
    g = torch.Generator(device=dev)
    g.manual_seed(300 + 1000 * member + tag)
    hdr = b"@NH1:7:HGF2YDSXX:L:TTTT:XXXXX:YYYYY M:N:0:GATTACAG\n"
    reclen = len(hdr) + L + 3 + L + 1
    rec = torch.empty((n, reclen), dtype=torch.uint8, device=dev)
    rec[:, :len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
    gi = torch.arange(n, device=dev, dtype=torch.int64) + member * n  # the same ids in both mate files
    gid = torch.Generator(device=dev)
    gid.manual_seed(77 + member)

    def digits(col, val, width):
        for d in range(width):
            rec[:, col + width - 1 - d] = (48 + (val // (10 ** d)) % 10).to(torch.uint8)

    digits(hdr.index(b"L:"), 1 + (gi // 12_500_000) % 4, 1)
    digits(hdr.index(b"TTTT"), 1101 + (gi // 50_000) % 1000, 4)
    digits(hdr.index(b"XXXXX"), 10_000 + torch.randint(0, 25_000, (n,), generator=gid, device=dev), 5)
    digits(hdr.index(b"YYYYY"), 10_000 + ((gi % 50_000) * 17) // 10, 5)
    rec[:, hdr.index(b" M:") + 1] = 48 + tag
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    p = len(hdr)
    rec[:, p:p + L] = acgt[torch.randint(0, 4, (n, L), generator=g, device=dev)]
    rec[:, p + L:p + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev)
    # qualities: 'F' up to a per-read point (6 % ':' sprinkled in), behind it a mix of ':' ',' '#'
    u = torch.rand((n, 1), generator=g, device=dev)
    decay = (L * (0.30 + 0.70 * u ** 0.4)).to(torch.int64)
    pos = torch.arange(L, device=dev, dtype=torch.int64)[None, :]
    r = torch.rand((n, L), generator=g, device=dev)
    q = torch.full((n, L), 70, dtype=torch.uint8, device=dev)           # 'F'
    q[(pos < decay) & (r < 0.06)] = 58                                  # ':'
    late = pos >= decay
    q[late & (r < 0.45)] = 58
    q[late & (r >= 0.45) & (r < 0.75)] = 44                             # ','
    q[late & (r >= 0.92)] = 35                                          # '#'
    rec[:, p + L + 3:p + 2 * L + 3] = q
    rec[:, p + 2 * L + 3] = 10
    rec.cpu().numpy().tofile(path)
    return n * reclen


def make_e2e_inputs(cx, args, tmp, n, L, reps, threads):
    """The gzip FASTQ pair of the e2e legs in `tmp`: two files of `reps` gzip members of n synthetic pairs each, the distinct
    members in rotation (A B C D E A B ...), level 6 by the library's block-parallel host encoder.  Returns the files and what
    the output checks need: every member's text length and xxh3-64."""
    import shutil
    import threading
    from nohuman_amd import _lib
    t0 = time.time()
    files = [os.path.join(tmp, "r_%d.fq.gz" % tag) for tag in (1, 2)]
    distinct = max(1, min(reps, args.e2e_distinct))
    dist_hash = {1: [None] * distinct, 2: [None] * distinct}
    dist_len = {1: [0] * distinct, 2: [0] * distinct}
    errors = []

    host_gz = []  # (bytes of text, seconds) of the host encoder at work on the inputs

    def compress_member(plain, tag, k):  # member k is compressed while member k+1 is generated
        try:
            dist_hash[tag][k] = _hash_file_ranges(plain, [(0, os.path.getsize(plain))])[0]
            tc = time.perf_counter()
            if _lib.lib().nh_compress_file(os.fsencode(plain), os.fsencode(plain + ".gz"), 2, max(1, threads // 2)) != 0:
                raise RuntimeError("nh_compress_file failed")
            host_gz.append((os.path.getsize(plain), time.perf_counter() - tc))
            os.remove(plain)
        except Exception as ex:  # surfaced after the join
            errors.append(ex)

    pending = []
    for k in range(distinct):
        for tag in (1, 2):
            plain = os.path.join(tmp, "m_%d_%d.fq" % (k, tag))
            dist_len[tag][k] = e2e_member(cx, n, L, tag, k, plain)
            while len(pending) >= 2:
                pending.pop(0).join()
            th = threading.Thread(target=compress_member, args=(plain, tag, k))
            th.start()
            pending.append(th)
    for th in pending:
        th.join()
    if errors:
        raise errors[0]
    # the input files: `reps` members, the distinct ones in rotation (A B C D E A B ...)
    member_hash = {tag: [dist_hash[tag][i % distinct] for i in range(reps)] for tag in (1, 2)}
    member_len = {tag: [dist_len[tag][i % distinct] for i in range(reps)] for tag in (1, 2)}
    for tag in (1, 2):
        with open(files[tag - 1], "wb") as out:
            for i in range(reps):
                with open(os.path.join(tmp, "m_%d_%d.fq.gz" % (i % distinct, tag)), "rb") as src:
                    shutil.copyfileobj(src, out, 16 << 20)
        for k in range(distinct):
            os.remove(os.path.join(tmp, "m_%d_%d.fq.gz" % (k, tag)))
    t_setup = time.time() - t0
    text_bytes = sum(member_len[1]) + sum(member_len[2])
    gz_bytes = sum(os.path.getsize(f) for f in files)
    return dict(files=files, member_len=member_len, member_hash=member_hash, text_bytes=text_bytes, gz_bytes=gz_bytes,
                t_setup=t_setup, host_gz=host_gz, distinct=distinct)


def e2e_multi_prepare(cx, args, eng):
    """Inputs and database directory (tmpfs) of the one-process multi-device leg: a cyclic configs[4]-shaped gzip pair of
    `--e2e-multi-pairs` pairs per device (scaled down to what the host's memory holds, scale stated), and the engine's table
    written out as hash.k2d / opts.k2d / taxo.k2d -- nh_run() loads the database into every device itself."""
    import shutil
    import tempfile
    from nohuman_amd.dist import usable_cpu_count
    N = cx.world
    n, L = args.e2e_pairs, args.read_len
    want = max(1, -(-N * args.e2e_multi_pairs // n))  # members per file
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    info = eng.info
    db_bytes = 4 * info.capacity + (1 << 20)
    try:
        free = shutil.disk_usage(base or tempfile.gettempdir()).free
        with open("/proc/meminfo") as f:
            avail = [int(x.split()[1]) * 1024 for x in f if x.startswith("MemAvailable:")][0]
        room = min(free, avail)
    except (OSError, IndexError, ValueError):
        room = 1 << 62
    # per member pair: gzip in + gzip out (4.1 : 1 each) and, while ONE output is checked, its text
    per_rep = n * 2 * (51 + 2 * L + 4) * (2 / 4.0) + n * (51 + 2 * L + 4)
    reps = want
    while reps > 1 and per_rep * reps + db_bytes + (16 << 30) > 0.6 * room:
        reps -= 1
    tmp = tempfile.mkdtemp(prefix="nh_bench_multi_", dir=base)
    try:
        threads = usable_cpu_count()
        inp = make_e2e_inputs(cx, args, tmp, n, L, reps, threads)
        db = os.path.join(tmp, "db")
        write_k2_db(db, eng.opts_image(), eng.taxonomy_image(), (info.capacity, info.size, info.key_bits, info.value_bits), eng.download_table())
    except Exception:
        shutil.rmtree(tmp, ignore_errors=True)
        raise
    return dict(tmp=tmp, db=db, files=inp["files"], member_len={str(k): v for k, v in inp["member_len"].items()},
                member_hash={str(k): v for k, v in inp["member_hash"].items()}, pairs=n * reps, wanted_pairs=n * want, devices=N,
                threads=threads, text_bytes=inp["text_bytes"], gz_bytes=inp["gz_bytes"], setup_s=round(inp["t_setup"], 1),
                fake_devices=bool(getattr(cx, "one_gpu", False)))


def e2e_multi_child(spec_path):
    """The fresh process of the multi-device leg (python bench.py --multi-child spec.json): nh_run(n_devices = N), gzip in ->
    gzip out, twice (the second run finds the process's buffers warm, like the single-device legs); one JSON line."""
    spec = json.load(open(spec_path))
    import nohuman_amd
    o1, o2 = os.path.join(spec["tmp"], "mo_1.fq.gz"), os.path.join(spec["tmp"], "mo_2.fq.gz")
    runs = []
    for k in range(2):
        for pth in (o1, o2):
            if os.path.exists(pth):
                os.remove(pth)
        tr_path = os.path.join(spec["tmp"], "multi_trace_%d.txt" % k)
        saved = os.dup(2)
        fd = os.open(tr_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        os.environ["NOHUMAN_TRACE"] = "1"
        try:
            os.dup2(fd, 2)
            t = time.perf_counter()
            st = nohuman_amd.engine.run(spec["db"], spec["files"][0], o1, in2=spec["files"][1], out2=o2, threads=spec["threads"],
                                        device_ids=list(range(spec["devices"])), out_codec=2, codec_threads=max(1, spec["threads"] // 2))
            dt = time.perf_counter() - t
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            os.close(fd)
        tr = open(tr_path).read()
        pick = lambda key: [ln.split("]", 1)[1].strip() for ln in tr.splitlines() if key in ln]  # noqa: E731
        runs.append({"wall_s": round(dt, 4), "run_s": round(st.seconds, 4), "total": int(st.total_sequences), "classified": int(st.classified),
                     "pieces_by_device": pick("pieces by device"), "counters": pick("counters reduced by"), "phases": pick("nh_run: database into HBM"),
                     "warn": [ln for ln in tr.splitlines() if "WARN" in ln or "DEVICE DISCIPLINE" in ln][:4]})
    print(json.dumps({"multi_child": runs}), flush=True)
    return 0


def e2e_multi_run(spec):
    """Starts the child, checks its outputs (each inflated by the library's host decoder, xxh3-64 per member == the generated
    text), and condenses what it printed: Mreads/s over the run's own seconds (database load excluded, as kraken2's timer),
    which device decoded how many pieces of each input, what reduced the counters."""
    import ctypes
    import re
    from nohuman_amd import _lib
    sp = os.path.join(spec["tmp"], "spec.json")
    with open(sp, "w") as fo:
        json.dump(spec, fo)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    if spec["fake_devices"]:  # NOHUMAN_BENCH_ONE_GPU: N logical devices on the one GPU (nh_internal.h)
        env["NOHUMAN_FAKE_DEVICES"] = str(spec["devices"])
    t = time.perf_counter()
    # (bounded: this leg has never met a real multi-GPU node -- a child that hangs there must not cost the scaling line more than minutes)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-child", sp], env=env, capture_output=True, text=True,
                       timeout=float(os.environ.get("NOHUMAN_BENCH_MULTI_TIMEOUT", "300")))
    t_child = time.perf_counter() - t
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"multi_child"')]
    if p.returncode != 0 or not lines:
        raise RuntimeError("multi-device child failed (%d): %s" % (p.returncode, (p.stderr or p.stdout)[-400:]))
    runs = json.loads(lines[-1])["multi_child"]
    best = min(runs, key=lambda r: r["run_s"])
    ok = {}
    threads = spec["threads"]
    for tag in ("1", "2"):  # one output at a time: its text is in tmpfs only while it is checked
        gzp = os.path.join(spec["tmp"], "mo_%s.fq.gz" % tag)
        plain = os.path.join(spec["tmp"], "mo_%s.fq" % tag)
        st3 = (ctypes.c_uint64 * 3)()
        rc = _lib.lib().nh_gunzip_file(os.fsencode(gzp), os.fsencode(plain), threads, 0, st3)
        os.remove(gzp)
        offs, pos = [], 0
        for ln in spec["member_len"][tag]:
            offs.append((pos, ln))
            pos += ln
        ok[tag] = rc == 0 and os.path.getsize(plain) == pos and _hash_file_ranges(plain, offs) == spec["member_hash"][tag]
        os.remove(plain)
    pieces = {}
    for ln in best["pieces_by_device"]:  # "gzip reader, <path>: pieces by device (device:pieces) 0:3 1:2"
        m = re.search(r"gzip reader, (\S+): pieces by device \(device:pieces\)(.*)", ln)
        if m:
            pieces[os.path.basename(m.group(1))] = m.group(2).strip()
    load = re.search(r"database into HBM ([0-9.]+) s", " ".join(best["phases"]))
    backend = (best["counters"] or ["host sums (no collective ran)"])[0].replace("counters reduced by ", "")
    nums = {
        "value": round(2 * best["total"] / best["run_s"] / 1e6, 3), "unit": "Mreads/s", "wall_s": best["run_s"], "call_wall_s": best["wall_s"],
        "db_load_s": float(load.group(1)) if load else None, "devices": spec["devices"],
        "logical_devices_on_one_gpu": bool(spec["fake_devices"]), "pairs": best["total"], "pairs_per_device": best["total"] // spec["devices"],
        "scale_of_request": round(spec["pairs"] / max(spec["wanted_pairs"], 1), 3), "host_threads": threads,
        "rccl_backend": backend[:120], "pieces_by_device": "; ".join("%s %s" % kv for kv in sorted(pieces.items()))[:200],
        "outputs_equal_inputs": bool(ok["1"] and ok["2"] and best["total"] == spec["pairs"] and best["classified"] == 0),
        "first_run_wall_s": runs[0]["run_s"], "warnings": len(best["warn"]),
    }
    details = {"what": "ONE process, nh_run(n_devices = %d), gzip in -> gzip out on a cyclic configs[4]-shaped pair (%d pairs, %.2f GB of gzip, "
                       "%.2f GB of text), database loaded into every device by the run itself; second of two runs in the process; child wall %.1f s, "
                       "inputs + database written in %.1f s" % (spec["devices"], spec["pairs"], spec["gz_bytes"] / 1e9, spec["text_bytes"] / 1e9, t_child, spec["setup_s"]),
               "runs": runs}
    return nums, details


def e2e_leg(cx, args, eng):
    """nh_run_engine on gzip FASTQ pairs: first byte read -> last byte written, database load excluded
    (as kraken2's own timer).  Inputs: `e2e_reps` DISTINCT gzip members of `e2e_pairs` synthetic pairs each
    per mate file (level 6, the library's own block-parallel encoder), Illumina-like ids and qualities, in
    tmpfs.  Every read is kept (random reads against a human table), so the outputs must be the input text
    byte for byte: checked member by member with xxh3-64, not by size."""
    import ctypes
    import shutil
    import tempfile
    import threading
    np, torch = cx.np, cx.torch
    from nohuman_amd import _lib
    from nohuman_amd.dist import usable_cpu_count
    threads = usable_cpu_count()
    n, L, reps = args.e2e_pairs, args.read_len, args.e2e_reps
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    # inputs and outputs live in memory (tmpfs): scale the leg down on a host that cannot hold them
    try:
        free = shutil.disk_usage(base or tempfile.gettempdir()).free
        with open("/proc/meminfo") as f:
            avail = [int(x.split()[1]) * 1024 for x in f if x.startswith("MemAvailable:")][0]
        room = min(free, avail)
    except (OSError, IndexError, ValueError):
        room = 1 << 62
    per_rep = n * 2 * (51 + 2 * L + 4) * 1.3  # text out + compressed in, per member pair
    while reps > 1 and per_rep * reps + (8 << 30) > 0.6 * room:
        reps -= 1
    if per_rep + (8 << 30) > 0.6 * room:
        return {"skipped": "not enough memory-backed space for the e2e inputs and outputs (%.1f GB usable)" % (room / 1e9)}
    tmp = tempfile.mkdtemp(prefix="nh_bench_e2e_", dir=base)
    try:
        inp = make_e2e_inputs(cx, args, tmp, n, L, reps, threads)
        files, member_len, member_hash, distinct = inp["files"], inp["member_len"], inp["member_hash"], inp["distinct"]
        text_bytes, gz_bytes, t_setup, host_gz = inp["text_bytes"], inp["gz_bytes"], inp["t_setup"], inp["host_gz"]
        o1, o2 = os.path.join(tmp, "o_1.fq"), os.path.join(tmp, "o_2.fq")
        best = None
        trace = ""
        for _ in range(2):  # first run warms the buffers (pinned allocations, page cache of the outputs)
            for p in (o1, o2):
                if os.path.exists(p):
                    os.remove(p)
            tr_path = os.path.join(tmp, "trace.txt")
            saved = os.dup(2)
            fd = os.open(tr_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
            os.environ["NOHUMAN_TRACE"] = "1"
            try:
                os.dup2(fd, 2)
                t = time.perf_counter()
                st = eng.run(files[0], o1, in2=files[1], out2=o2, threads=threads)
                dt = time.perf_counter() - t
            finally:
                os.dup2(saved, 2)
                os.close(saved)
                os.close(fd)
                os.environ.pop("NOHUMAN_TRACE", None)
            if best is None or dt < best[0]:
                best = (dt, st.total_sequences, st.classified)
                trace = " | ".join(x.strip() for x in open(tr_path).read().strip().splitlines()
                                   if "wall" in x or "gunzip consumer" in x or "gzip reader" in x)
        dt, nfr, ncl = best
        # every read was kept: the outputs are the generated text, member by member (byte-exact, by digest)
        ok = nfr == n * reps and ncl == 0
        checked = {}

        def verify(tag, path):
            offs, pos = [], 0
            for ln in member_len[tag]:
                offs.append((pos, ln))
                pos += ln
            good = os.path.getsize(path) == pos and _hash_file_ranges(path, offs) == member_hash[tag]
            checked[tag] = good

        vt = [threading.Thread(target=verify, args=(1, o1)), threading.Thread(target=verify, args=(2, o2))]
        tv = time.perf_counter()
        for th in vt:
            th.start()
        for th in vt:
            th.join()
        t_verify = time.perf_counter() - tv
        ok = ok and checked.get(1) is True and checked.get(2) is True
        for p in (o1, o2):
            os.remove(p)
        def which_reader(tr):  # "[nohuman trace] gzip reader: GPU / GPU" -> what nh_run chose (device_reader_pays, nh_run.hip)
            for ln in tr.splitlines():
                if "gzip reader:" in ln:
                    return ln.split("gzip reader:")[1].strip()
            return "?"

        def traced_run(label, **kw):
            tr_path = os.path.join(tmp, "trace_%s.txt" % label)
            saved = os.dup(2)
            fd = os.open(tr_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
            os.environ["NOHUMAN_TRACE"] = "1"
            try:
                os.dup2(fd, 2)
                t = time.perf_counter()
                st = eng.run(files[0], kw.pop("o1"), in2=files[1], out2=kw.pop("o2"), threads=threads, **kw)
                dt = time.perf_counter() - t
            finally:
                os.dup2(saved, 2)
                os.close(saved)
                os.close(fd)
                os.environ.pop("NOHUMAN_TRACE", None)
            return st, dt, open(tr_path).read()

        reader_plain = which_reader(open(os.path.join(tmp, "trace.txt")).read())
        # the input side alone: same run with --classified-out semantics (nothing is kept, nothing written)
        st_in, dt_in, tr_in = traced_run("none", o1=os.path.join(tmp, "h_1.fq"), o2=os.path.join(tmp, "h_2.fq"), keep_human=True)
        # the same run with gzip outputs, the reference's default for gzip inputs (main.rs:238-245): the kept reads
        # are compressed on the GPU (nh_deflate.hip) as the writer hands them over; the files are inflated again
        # by the library's own reader and compared with the generated text like the plain outputs
        gz = {}
        gz_detail = {}
        try:
            g1, g2 = os.path.join(tmp, "o_1.fq.gz"), os.path.join(tmp, "o_2.fq.gz")
            st_gz, dt_gz, tr_gz = traced_run("gz", o1=g1, o2=g2, out_codec=2, codec_threads=max(1, threads // 2))
            out_gz = os.path.getsize(g1) + os.path.getsize(g2)
            gz_ok = {}

            def verify_gz(tag, gzpath, plain):
                st3 = (ctypes.c_uint64 * 3)()
                rc = _lib.lib().nh_gunzip_file(os.fsencode(gzpath), os.fsencode(plain), max(1, threads // 2), 0, st3)
                os.remove(gzpath)
                offs, pos = [], 0
                for ln in member_len[tag]:
                    offs.append((pos, ln))
                    pos += ln
                gz_ok[tag] = rc == 0 and os.path.getsize(plain) == pos and _hash_file_ranges(plain, offs) == member_hash[tag]
                os.remove(plain)

            vt = [threading.Thread(target=verify_gz, args=(1, g1, o1)), threading.Thread(target=verify_gz, args=(2, g2, o2))]
            for th in vt:
                th.start()
            for th in vt:
                th.join()
            host_rate = sum(b for b, _ in host_gz) / max(1e-9, sum(sec for _, sec in host_gz)) if host_gz else 0.0
            gz = {"value": round(2 * st_gz.total_sequences / dt_gz / 1e6, 3), "wall_s": round(dt_gz, 4), "reader": which_reader(tr_gz),
                  "output_ratio": round(text_bytes / max(out_gz, 1), 3),
                  "outputs_equal_inputs": bool(gz_ok.get(1) is True and gz_ok.get(2) is True and st_gz.total_sequences == n * reps),
                  "host_encoder_GBps": round(host_rate / 1e9, 3)}
            gz_detail = {"what": "same inputs, outputs written as gzip (out_codec 2): one ordinary member per file, encoded on the GPU, "
                                 "64 KiB of text per wave; both files inflated by nh_gunzip_file, xxh3-64 per member range == the generated text",
                         "stages": " | ".join(x.strip() for x in tr_gz.splitlines() if "wall" in x or "gzip encoder" in x or "gzip reader" in x),
                         "host_encoder_what": "nh_compress_file (zlib -6 blocks on %d workers, what gzp does in the reference) on this "
                                              "leg's own input members, two files at a time" % max(1, threads // 2)}
        except Exception as ex:  # reported, not fatal for the line
            gz = {"error": str(ex)[:300]}
        # each reader by NAME on the same inputs (NOHUMAN_GZ_READER=device: nh_gunzip.hip on the GPU; =host: nh_inflate.cpp on
        # the host cores): what the choice above was made from
        named = {}
        keep_env = os.environ.get("NOHUMAN_GZ_READER")
        try:
            for reader in ("device", "host"):
                os.environ["NOHUMAN_GZ_READER"] = reader
                t = time.perf_counter()
                st_o = eng.run(files[0], os.path.join(tmp, "h_1.fq"), in2=files[1], out2=os.path.join(tmp, "h_2.fq"),
                               threads=threads, keep_human=True)
                dt_o = time.perf_counter() - t
                t = time.perf_counter()
                st_og = eng.run(files[0], os.path.join(tmp, "x_1.fq.gz"), in2=files[1], out2=os.path.join(tmp, "x_2.fq.gz"),
                                threads=threads, out_codec=2, codec_threads=max(1, threads // 2))
                dt_og = time.perf_counter() - t
                for pth in ("x_1.fq.gz", "x_2.fq.gz"):
                    os.remove(os.path.join(tmp, pth))
                for p in (o1, o2):
                    if os.path.exists(p):
                        os.remove(p)
                t = time.perf_counter()
                st_op = eng.run(files[0], o1, in2=files[1], out2=o2, threads=threads)
                dt_op = time.perf_counter() - t
                named[reader] = {"input_side_only": round(2 * st_o.total_sequences / dt_o / 1e6, 3),
                                 "gzip_to_gzip": round(2 * st_og.total_sequences / dt_og / 1e6, 3),
                                 "gzip_to_plain": round(2 * st_op.total_sequences / dt_op / 1e6, 3)}
        except Exception as ex:
            named["error"] = str(ex)[:300]
        finally:
            if keep_env is None:
                os.environ.pop("NOHUMAN_GZ_READER", None)
            else:
                os.environ["NOHUMAN_GZ_READER"] = keep_env
        nums = {
            "unit": "Mreads/s",
            "pairs": int(nfr),
            "scale_of_configs2": round(nfr / 50e6, 3),
            "host_threads": threads,
            "gzip_ratio": round(text_bytes / max(gz_bytes, 1), 2),
            "gzip_to_plain": {"value": round(2 * nfr / dt / 1e6, 3), "wall_s": round(dt, 4), "reader": reader_plain,
                              "outputs_equal_inputs": bool(ok)},
            "gzip_to_gzip": gz,
            "input_side_only": {"value": round(2 * st_in.total_sequences / dt_in / 1e6, 3), "wall_s": round(dt_in, 4),
                                "reader": which_reader(tr_in)},
            "outputs_equal_inputs": bool(ok and gz.get("outputs_equal_inputs") is True),
            "readers_by_name": named,
            "setup_seconds": round(t_setup, 1),
        }
        details = {
            "workload": "%d pairs of %d bp = 2 gzip FASTQ files of %d members x %d pairs (%d distinct members in rotation; "
                        "level 6; Illumina-style ids, binned qualities; %.2f GB compressed, %.2f GB of text: %.2f:1), every read "
                        "kept; configs[2] shape at %.0f %% scale"
                        % (nfr, L, reps, n, distinct, gz_bytes / 1e9, text_bytes / 1e9, text_bytes / max(gz_bytes, 1), 100.0 * nfr / 50e6),
            "outputs_check": "xxh3-64 of every member's byte range of both outputs == xxh3-64 of the generated text (%.1f s)" % t_verify,
            "gzip_to_plain_stages": trace,
            "input_side_only": "same inputs, keep_human=1 (no read is kept: inflate + parse + H2D + classify + D2H, no output bytes)",
            "gzip_to_gzip": gz_detail,
            "reader": "chosen by nh_run (device_reader_pays, nh_run.hip) unless named; 'GPU' = nh_gunzip.hip (block search, decode, window "
                      "scan, marker resolve, CRC-32 and the record index on the GPU), 'host' = nh_inflate.cpp + nh_fastx.cpp on the host cores",
        }
        return nums, details
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_ont_input(cx, tmp, n_member, members, threads):
    """ONE gzip FASTQ file of `members` x `n_member` ONT-like reads (length lognormal(8.8, 0.85) in [200, 200000]) under tmp:
    (path, text bytes of a member, xxh3-64 of a member's text, N50).  Used by e2e_ont_leg and tools/ont_trace.py."""
    import shutil
    np, torch = cx.np, cx.torch
    from nohuman_amd import _lib
    dev = cx.dev
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    lens = torch.exp(torch.randn(n_member, generator=g, device=dev, dtype=torch.float64) * 0.85 + 8.8).clamp(200, 200000).to(torch.int64)
    total = int(lens.sum().item())
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    seq = acgt[torch.randint(0, 4, (total,), generator=g, device=dev)].cpu().numpy()
    lens_h = lens.cpu().numpy()
    srt = np.sort(lens_h)[::-1]
    n50 = int(srt[np.searchsorted(np.cumsum(srt), total / 2)])
    plain = os.path.join(tmp, "ont.fq")
    with open(plain, "wb") as fo:
        off = 0
        rows = []
        for i in range(n_member):
            ln = int(lens_h[i])
            rows.append(b"@ont.%d runid=nh ch=%d\n" % (i, i % 512))
            rows.append(seq[off:off + ln].tobytes())
            rows.append(b"\n+\n" + b"5" * ln + b"\n")
            off += ln
            if len(rows) >= 3000:
                fo.write(b"".join(rows))
                rows = []
        fo.write(b"".join(rows))
    del seq
    text_len = os.path.getsize(plain)
    member_hash = _hash_file_ranges(plain, [(0, text_len)])[0]
    if _lib.lib().nh_compress_file(os.fsencode(plain), os.fsencode(plain + ".gz"), 2, threads) != 0:
        raise RuntimeError("nh_compress_file failed")
    os.remove(plain)
    fin = os.path.join(tmp, "ont_all.fq.gz")
    with open(fin, "wb") as out:
        for _ in range(members):
            with open(plain + ".gz", "rb") as src:
                shutil.copyfileobj(src, out, 16 << 20)
    os.remove(plain + ".gz")
    return fin, text_len, member_hash, n50


def e2e_ont_leg(cx, args, eng):
    """configs[3] end to end, scaled: ONE gzip FASTQ file of ONT-like reads (length lognormal(8.8, 0.85) in [200, 200000],
    N50 ~ 10 kb) through nh_run, gzip in -> gzip out (the reference's default for a gzip input, main.rs:238-245) and the input
    side alone; the output is inflated again and compared with the generated text by xxh3-64 per member."""
    import ctypes
    import shutil
    import tempfile
    np, torch = cx.np, cx.torch
    from nohuman_amd import _lib
    from nohuman_amd.dist import usable_cpu_count
    threads = usable_cpu_count()
    n_member = int(os.environ.get("NOHUMAN_BENCH_ONT_READS", "100000"))
    members = int(os.environ.get("NOHUMAN_BENCH_ONT_MEMBERS", "5"))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="nh_bench_ont_", dir=base)
    try:
        t0 = time.time()
        fin, text_len, member_hash, n50 = make_ont_input(cx, tmp, n_member, members, threads)
        t_setup = time.time() - t0
        gout = os.path.join(tmp, "o.fq.gz")
        best = None
        for _ in range(2):
            if os.path.exists(gout):
                os.remove(gout)
            t = time.perf_counter()
            st = eng.run(fin, gout, threads=threads, out_codec=2, codec_threads=max(1, threads // 2))
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, st.total_sequences, st.classified, st.total_bases)
        dt, nfr, ncl, nb = best
        back = os.path.join(tmp, "back.fq")
        st3 = (ctypes.c_uint64 * 3)()
        rc = _lib.lib().nh_gunzip_file(os.fsencode(gout), os.fsencode(back), threads, 0, st3)
        same = rc == 0 and os.path.getsize(back) == text_len * members and \
            _hash_file_ranges(back, [(k * text_len, text_len) for k in range(members)]) == [member_hash] * members
        out_ratio = text_len * members / max(os.path.getsize(gout), 1)
        os.remove(back)
        os.remove(gout)
        t = time.perf_counter()
        st_in = eng.run(fin, os.path.join(tmp, "h.fq"), threads=threads, keep_human=True)
        dt_in = time.perf_counter() - t
        named = {}
        keep_env = os.environ.get("NOHUMAN_GZ_READER")
        try:
            for reader in ("device", "host"):
                os.environ["NOHUMAN_GZ_READER"] = reader
                t = time.perf_counter()
                eng.run(fin, gout, threads=threads, out_codec=2, codec_threads=max(1, threads // 2))
                named[reader] = round(nfr / (time.perf_counter() - t) / 1e6, 4)
                os.remove(gout)
        finally:
            if keep_env is None:
                os.environ.pop("NOHUMAN_GZ_READER", None)
            else:
                os.environ["NOHUMAN_GZ_READER"] = keep_env
        nums = {"ont_gzip_to_gzip": {"value": round(nfr / dt / 1e6, 4), "wall_s": round(dt, 4), "Gbases_per_s": round(nb / dt / 1e9, 3),
                                     "reads": int(nfr), "N50": n50, "outputs_equal_inputs": bool(same and nfr == n_member * members and ncl == 0),
                                     "output_ratio": round(out_ratio, 3), "readers_by_name": named},
                "ont_input_side_only": {"value": round(st_in.total_sequences / dt_in / 1e6, 4), "wall_s": round(dt_in, 4)}}
        details = {"workload": "%d ONT-like reads = ONE gzip FASTQ file of %d members x %d reads (the same member %d times; level 6; %.2f GB "
                               "compressed, %.2f GB of text), %.2f Gbases, N50 %d; configs[3] shape at %.1f %% scale; every read kept, gzip out; "
                               "output inflated by nh_gunzip_file, xxh3-64 per member == the generated text"
                               % (nfr, members, n_member, members, os.path.getsize(fin) / 1e9, text_len * members / 1e9, nb / 1e9, n50, 100.0 * nfr / 10e6),
                   "setup_seconds": round(t_setup, 1)}
        return nums, details
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
