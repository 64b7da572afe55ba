"""Row (e) of SURVEY §8 on CPU: 2-rank gloo run of the sharding + cost all-gather logic (no GPU compute here)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import numpy as np

from adaptigraph_amd.sharding import shard_bounds, shard_bounds_weighted, all_gather_costs, sharded_rollout_costs


def test_shard_bounds_cover_batch():
    for B in (1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_weighted_shards_balance_work():
    """cfg 5 (ragged batch): contiguous shards balanced by a per-candidate work estimate, identical on every rank."""
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for trial in range(5):
            w = rng.integers(300, 2026, size=int(rng.integers(world, 600))).astype(np.float64) * rng.choice([6, 11, 21])
            spans = [shard_bounds_weighted(w, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == len(w)
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            loads = [w[lo:hi].sum() for lo, hi in spans]
            assert max(loads) - min(loads) <= 2 * w.max() + 1e-9       # within one candidate of the ideal cut each side
    # a heavy head: the count-balanced split would give rank 0 most of the work
    w = np.r_[np.full(10, 100.0), np.full(90, 1.0)]
    assert shard_bounds_weighted(w, 2, 0) == (0, 5) and shard_bounds_weighted(w, 2, 1) == (5, 100)
    assert shard_bounds_weighted(np.zeros(4), 2, 0)[0] == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # every rank draws the same action batch
    actions = torch.rand(B, 2, 4)

    def fake_rollout(a):                                   # stands in for dynamics(): per-candidate, deterministic
        return a.sum(-1, keepdim=True).repeat(1, 1, 5)     # (b, H, 5)

    def cost(seq):
        return seq[:, -1].mean(-1)

    full = sharded_rollout_costs(fake_rollout, cost, actions)
    want = cost(fake_rollout(actions))
    ok = torch.equal(full, want)                           # sharded == unsharded, bit for bit
    lo, hi = shard_bounds(B, world, rank)
    ok = ok and torch.equal(all_gather_costs(want[lo:hi], B), want)
    # batch-global maxima of the cost functions (plan.py:37, losses.py:62) under sharding
    from adaptigraph_amd.losses import _global_max
    ok = ok and float(_global_max(want[lo:hi], True)) == float(want.max())
    # ragged batch: work-balanced shards, variable-length pieces through the same all-gather
    weights = torch.arange(B, dtype=torch.float64).numpy() ** 2 + 1.0
    full_w = sharded_rollout_costs(lambda a, lo, hi: fake_rollout(a), lambda seq, lo, hi: cost(seq), actions, weights=weights)
    ok = ok and torch.equal(full_w, want)
    blo, bhi = shard_bounds_weighted(weights, world, rank)
    ok = ok and (bhi - blo) != (hi - lo)                   # really a different split than the count-balanced one
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_unsharded():
    ctx = mp.get_context("spawn")
    for B in (64, 65):                                     # even and ragged split
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert sorted(res) == [(0, True), (1, True)]
