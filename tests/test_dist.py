"""N>1 path on CPU: world_size-2 gloo run of the shard plan + the counter all-reduce (the only
collective of the path; RCCL on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nohuman_amd.dist import reduce_counters, shard_range, usable_cpu_count


def test_shard_range_partitions_in_order():
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_usable_cpu_count_positive():
    assert usable_cpu_count() >= 1


def _worker(rank, world, port, n_frag, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frag, rank, world)
    # per-rank counters as the engine would report them for its shard:
    # {fragments, classified (every 3rd fragment), bases (150 per fragment), lookups (39 each)}
    frs = torch.arange(lo, hi)
    counters = torch.tensor([hi - lo, int((frs % 3 == 0).sum()), 150 * (hi - lo), 39 * (hi - lo)],
                            dtype=torch.int64)
    tot, tmax = reduce_counters(counters, 1.0 + rank)
    q.put((rank, tot, tmax))
    dist.destroy_process_group()


def test_counter_all_reduce_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frag = 1001
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frag, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [n_frag, len([i for i in range(n_frag) if i % 3 == 0]), 150 * n_frag, 39 * n_frag]
    for rank, tot, tmax in out:
        assert tot == want
        assert tmax == 2.0  # the slowest rank defines the job time


def _shard_worker(rank, world, port, paired, q):
    """One rank of the N>1 path on CPU: classify this rank's contiguous shard (the oracle stands in
    for the per-rank GPU engine), then the path's only collective on the real counters."""
    import numpy as np
    from oracle import oracle as orc
    from tests import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ob, tb, hb, genomes, _ = synth.toy_db()  # DB replicated: every rank builds the same images
    reads = synth.sample_reads(np.random.default_rng(11), genomes, 1001, paired=paired, len_jitter=30)
    lo, hi = shard_range(len(reads), rank, world)
    bases, offs = orc.pack_reads(reads[lo:hi], paired)
    out, lookups = orc.OracleDB(ob, tb, hb).classify(bases, offs, paired, 0.05)
    counters = torch.tensor([hi - lo, int((out["call"] != 0).sum()), int(offs[-1]), int(lookups.sum())],
                            dtype=torch.int64)
    tot, tmax = reduce_counters(counters, 0.25 * (rank + 1))
    q.put((rank, lo, hi, out.tobytes(), tot, tmax))
    dist.destroy_process_group()


@pytest.mark.parametrize("paired", [False, True])
def test_real_shards_reduce_to_unsharded_totals(paired):
    import numpy as np
    from oracle import oracle as orc
    from tests import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, paired, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the unsharded run
    ob, tb, hb, genomes, _ = synth.toy_db()
    reads = synth.sample_reads(np.random.default_rng(11), genomes, 1001, paired=paired, len_jitter=30)
    bases, offs = orc.pack_reads(reads, paired)
    out, lookups = orc.OracleDB(ob, tb, hb).classify(bases, offs, paired, 0.05)
    want = [len(reads), int((out["call"] != 0).sum()), int(offs[-1]), int(lookups.sum())]
    assert want[1] > 100  # the shards really classify something
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == len(reads)  # contiguous, in order
    assert b"".join(g[3] for g in got) == out.tobytes()  # rank-order concatenation == input order
    for _, _, _, _, tot, tmax in got:
        assert tot == want and tmax == 0.5


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus N` started directly spawns its own ranks; the
    dry-run switch swaps the GPU work for one gloo all-reduce so that the plumbing runs on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    env = dict(os.environ, NOHUMAN_BENCH_DRYRUN="1", NOHUMAN_BENCH_LOGDIR=tempfile.mkdtemp())
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_counted"] == 2 and rec["world_size_seen"] == 2


@pytest.mark.parametrize("how", ["self-launch", "torch.distributed.run"])
def test_bench_eight_ranks_dry_run(tmp_path, how):
    """configs[4] is 8 GPUs of one node and no such node has been available (SCALE skipped every round): the launch plumbing
    of `bench.py --gpus 8` at its real width on CPU -- both ways it is started: directly (it spawns its ranks) and the
    driver's way, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ...`."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NOHUMAN_BENCH_DRYRUN="1", NOHUMAN_BENCH_LOGDIR=str(tmp_path), OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if how == "self-launch":
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"]
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_counted"] == 8 and rec["world_size_seen"] == 8


def test_bench_launcher_names_a_dead_rank_and_returns_within_seconds(tmp_path):
    """a rank that dies before the rendezvous used to leave rank 0 in init_process_group until torch's
    timeout.  The launcher polls all ranks, names the first non-zero exit, terminates the others and returns."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NOHUMAN_BENCH_DRYRUN="1", NOHUMAN_BENCH_FAIL_RANK="1", NOHUMAN_BENCH_LOGDIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    assert "rank 1 of 2 exited with status 7" in r.stderr, r.stderr[-1500:]
    assert took < 90, took  # (the first `import torch` of a fresh box alone can take a minute; the wait itself is < 1 s)
    assert os.path.exists(tmp_path / "bench_rank0.log") and os.path.exists(tmp_path / "bench_rank1.log")


def test_reduce_without_process_group_is_identity():
    tot, tmax = reduce_counters(torch.tensor([1, 2, 3, 4], dtype=torch.int64), 0.5)
    assert tot == [1, 2, 3, 4] and tmax == 0.5


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py has never met a multi-GPU node (SCALE skipped twice).  Short of one: two rank
    PROCESSES sharing GPU 0 (NOHUMAN_BENCH_ONE_GPU=1: RCCL refuses two ranks on one device, so the counters are
    reduced over gloo) -- self-launch, rendezvous, per-rank sharded batches through the HIP path, barriers,
    max-over-ranks timing, the counter reduction and the rank table are the real thing.  The line must count
    both ranks' fragments and look-ups."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NOHUMAN_BENCH_ONE_GPU="1", NOHUMAN_BENCH_LOGDIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--pairs", "200000",
           "--capacity", "200000033", "--pool", "2", "--no-cpu-baseline", "--no-e2e", "--no-variants"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size_seen"] == 2
    assert "gloo" in line["config"]["collective_backend"]
    ranks = line["config"]["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1] and all(x["kernel_ms"] > 0 for x in ranks)
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 2
    # 2 ranks x 4 steps x 200000 pairs, 38.7 look-ups per read: value = reads of BOTH ranks over the slower rank's time
    assert 37 < line["config"]["lookups_per_read"] < 40
    per_rank_reads_per_s = 200000 * 2 * 4 / (line["ms_per_step"] * 4 / 1e3)
    assert abs(line["value"] * 1e6 - 2 * per_rank_reads_per_s) / (2 * per_rank_reads_per_s) < 0.02


@pytest.mark.gpu
def test_bench_gpus_n_measures_the_products_own_multi_device_run(tmp_path):
    """`bench.py --gpus N` runs N single-device rank processes -- the product's own G > 1 path (ONE process,
    nh_run(n_devices = N): reader lanes over the devices, peer copies, the count all-reduce inside the library, one writer per
    file) had no bench leg, so an 8-GPU node would not have measured it.  Now rank 0, once every rank has closed its engine and
    left the process group, starts ONE fresh child that runs it on a cyclic configs[4]-shaped gzip pair.  Here: two ranks on
    the one GPU (NOHUMAN_BENCH_ONE_GPU), the child under NOHUMAN_FAKE_DEVICES=2 (two logical devices on the one GPU, the
    device discipline checked), at toy sizes with the reader's pieces scaled down so that both devices decode some."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NOHUMAN_BENCH_ONE_GPU="1", NOHUMAN_BENCH_LOGDIR=str(tmp_path), NOHUMAN_DEBUG_DEVICE="1",
               NOHUMAN_GZDEV_SEG="65536", NOHUMAN_GZDEV_STRETCH="4096", NOHUMAN_GZDEV_ROOM=str(8 << 20), NOHUMAN_BATCH_FRAGS="4096",
               NOHUMAN_GZDEV_MIN_BYTES="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "100000",
           "--capacity", "200000033", "--pool", "2", "--no-cpu-baseline", "--no-variants", "--e2e-pairs", "20000", "--e2e-distinct", "2",
           "--e2e-multi-pairs", "40000", "--wake-ms", "10"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    m = line["config"]["e2e"]["multi"]
    assert "error" not in m, m
    assert m["devices"] == 2 and m["logical_devices_on_one_gpu"] is True and m["pairs"] == 80000 and m["pairs_per_device"] == 40000
    assert m["outputs_equal_inputs"] is True and m["value"] > 0 and m["scale_of_request"] == 1.0
    # both logical devices decoded pieces of both inputs ("r_1.fq.gz 0:k 1:m; r_2.fq.gz ...")
    assert "r_1.fq.gz" in m["pieces_by_device"] and " 1:" in m["pieces_by_device"] and "0:" in m["pieces_by_device"], m["pieces_by_device"]
    c = line["config"]
    assert c["e2e_multi_value"] == m["value"] and c["e2e_multi_devices"] == 2 and c["e2e_multi_outputs_equal_inputs"] is True
    assert isinstance(c["e2e_multi_rccl_backend"], str) and isinstance(c["e2e_multi_pieces_by_device"], str)
