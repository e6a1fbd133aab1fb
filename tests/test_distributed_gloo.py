"""world_size-2 run of the N > 1 path on CPU (gloo): song sharding, the all-gather of force
vectors and the row-block split of the distance matrix — the same code bench.py drives with
backend "nccl" (RCCL) on GPUs.  The per-song analysis itself is GPU-only and is replaced
here by a deterministic stand-in vector per global song index."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bliss_amd.dist import gather_force_vectors, lpt_shards, shard_range


def _vec(i):
    rng = np.random.default_rng(1000 + i)
    return (rng.standard_normal(4) * 10).astype(np.float32)


def _dist_rows(all_v, first, count):
    d = all_v[first:first + count, None, :] - all_v[None, :, :]
    s = d[..., 0] * d[..., 0]
    for k in (1, 2, 3):
        s = (s + d[..., k] * d[..., k]).astype(np.float32)
    return np.sqrt(s).astype(np.float32)


def _worker(rank, world, port, total, ragged, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if ragged:
            lengths = [1000 + 37 * ((i * 7919) % 101) for i in range(total)]
            mine = lpt_shards(lengths, world)[rank]
            counts = [len(s) for s in lpt_shards(lengths, world)]
        else:
            first, count = shard_range(total, rank, world)
            mine = list(range(first, first + count))
            counts = [shard_range(total, r, world)[1] for r in range(world)]
        local = torch.from_numpy(np.stack([_vec(i) for i in mine]))
        allv = gather_force_vectors(local, counts).numpy()
        # every rank must hold every song's vector, rank-major
        order = (sum(lpt_shards(lengths, world), []) if ragged else list(range(total)))
        want = np.stack([_vec(i) for i in order])
        assert np.array_equal(allv, want), "all-gather result differs"
        my_first = sum(counts[:rank])
        rows = _dist_rows(allv, my_first, counts[rank])
        q.put((rank, my_first, rows, order))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(total, ragged):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    order = got[0][3]
    full = np.concatenate([g[2] for g in got], axis=0)
    allv = np.stack([_vec(i) for i in order])
    assert np.array_equal(full, _dist_rows(allv, 0, total))
    assert np.array_equal(full, full.T) and not np.any(np.diag(full))


def test_equal_shards_all_gather_and_row_blocks():
    _run(total=10, ragged=False)


def test_odd_total_and_lpt_shards():
    _run(total=11, ragged=False)
    _run(total=13, ragged=True)


def test_shard_helpers():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sum(c for _, c in (shard_range(65536, r, 8) for r in range(8))) == 65536
    lengths = [600, 10, 300, 300, 50, 590, 20, 20]
    shards = lpt_shards(lengths, 2)
    assert sorted(sum(shards, [])) == list(range(8))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 40
    for s in shards:
        assert [lengths[i] for i in s] == sorted((lengths[i] for i in s), reverse=True)


def test_bench_self_launch_plumbing():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks (torch.distributed.run
    on 127.0.0.1) and rank 0 prints exactly one JSON line; --plumbing-only runs that path on CPU:
    gloo group, shard ranges, the all-gather of stand-in vectors — the analysis itself is GPU-only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--plumbing-only", "--songs-per-gpu", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["results_ok"] is True
    assert line["collective"] == {"backend": "gloo", "all_gather_calls": 3, "bytes_per_rank": 80}
    assert line["config"]["parallelism"] == "shard2"
    assert len(line["per_rank"]["ms_per_step"]) == 2


def test_bench_plumbing_at_the_world_size_of_configs2():
    """The 8-rank job of BASELINE configs[2], as far as a box without GPUs can run it: `bench.py --gpus 8
    --plumbing-only` starts its own eight ranks, the gloo group forms, every rank contributes its shard and rank 0
    prints one line with the eight ranks' own clocks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--plumbing-only", "--songs-per-gpu", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["results_ok"] is True
    assert line["collective"] == {"backend": "gloo", "all_gather_calls": 3, "bytes_per_rank": 48}
    assert line["config"]["parallelism"] == "shard8" and "24 total" in line["config"]["workload"]
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and pr["ms_per_step_min"] <= pr["ms_per_step_max"]


def test_bench_refuses_more_gpus_than_devices():
    """Plain `--gpus 2` on a box with fewer devices fails before anything is launched, with a
    message about the device count (not about WORLD_SIZE)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "HIP device(s)" in r.stderr and "WORLD_SIZE" not in r.stderr
