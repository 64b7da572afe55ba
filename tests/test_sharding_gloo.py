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


def test_four_rank_gloo_gather_matches_unsharded():
    """world_size 4 (the 8-GPU node's collectives at half size): uneven shards 17/17/16/16, work-balanced ragged shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 66, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(4)]


# ------------------------------------------------------------------------------------------------- bench.py's step, 2 ranks
def _mpc_step_worker(rank, world, port, B, q):
    """The function bench.py times (sharding.sharded_candidate_rewards) with the engine replaced by its CPU oracle:
    actions -> contiguous shard -> rollout -> running_cost with its two batch-global maxima all-reduced -> all-gather ->
    every rank holds the same reward vector and picks the same best candidate as an unsharded evaluation."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import adaptigraph_oracle as O
    from oracle import costs_oracle as Cc
    from adaptigraph_amd.sharding import sharded_candidate_rewards
    from adaptigraph_amd.losses import _global_max
    rng = np.random.default_rng(3)                        # every rank builds the same inputs
    W = O.random_weights(3)
    task = dict(adj_thresh=0.75, topk=5, connect_tools_all=True, push_length=0.1, gripper_enable=True, eef_num=1,
                pusher_points=[[0.0, 0.0, 0.17]], max_nR=4000, sim_real_ratio=10, n_his=4)
    g = (np.arange(6) - 2.5) * 0.3
    xx, zz = np.meshgrid(g, g, indexing="ij")
    cloud = (np.stack([xx.ravel(), np.zeros(36), zz.ravel()], 1) + rng.normal(0, 0.02, (36, 3))).astype(np.float32)
    target = (cloud + np.float32([0.3, 0, 0.2])).astype(np.float32)
    acts = np.zeros((B, 2, 4), np.float32)
    acts[..., 0] = rng.uniform(-0.8, 0.8, (B, 2))
    acts[..., 1] = rng.uniform(-0.8, 0.8, (B, 2))
    acts[..., 2] = rng.uniform(-3, 3, (B, 2))
    acts[..., 3] = rng.integers(1, 3, (B, 2)) + 0.5
    actions = torch.from_numpy(acts)

    def rollout(a):
        return torch.from_numpy(O.dynamics(W, 3, cloud, a.numpy(), task)["state_seqs"])

    def make_reward(group):
        def reward(seq, a):                               # running_cost (plan.py:27-59) with the cloth objective
            s = seq.numpy()
            b, H = s.shape[:2]
            err = torch.from_numpy(Cc.chamfer(s.reshape(b * H, -1, 3), target[None]).reshape(b, H).astype(np.float32))
            w = (2.0 / (_global_max(err, group).to(torch.float64) + 1e-6)).to(torch.float32)          # plan.py:37
            raw = torch.from_numpy(Cc.cloth_penalty_terms(s, a.numpy(), cloud, 10.0))                 # (b,H,2)
            pen = 1.0 - raw[..., 0] - raw[..., 1] / _global_max(raw[..., 1], group) * 0.2              # losses.py:62-63
            return -w * err[:, -1] - 5.0 * pen.mean(1)
        return reward

    full = sharded_candidate_rewards(actions, rollout, make_reward(True))
    want = make_reward(None)(rollout(actions), actions)   # unsharded, on this rank alone
    ok = full.shape == (B,) and torch.equal(full, want) and int(torch.argmax(full)) == int(torch.argmax(want))
    q.put((rank, bool(ok), int(torch.argmax(full))))
    dist.destroy_process_group()


def test_two_rank_gloo_mpc_step_matches_unsharded():
    ctx = mp.get_context("spawn")
    B = 7                                                 # uneven split: 4 + 3 candidates
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mpc_step_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)] and res[0][2] == res[1][2]


# ------------------------------------------------------------------------------------- work-balanced shards (r05)
# With the contact-free prefix a candidate's work on the engine is 0 .. action_repeat forwards, depending on when its tool first
# touches the object.  Shards cut by candidate COUNT can then differ several-fold in work; sharding.work_balanced_bounds cuts them
# by forwards left (adaptigraph_amd.rollout_work on the GPU; here the oracle's restatement of it).  The reference has no multi-GPU
# code (plan.py:87: one device); its sampler is what makes the imbalance (plan_utils.py:48-50: uniform over the action box).
def _rope_case(n, rng):
    t = np.linspace(0, 1, n)
    p = np.stack([-3.5 + 2.5 * t, 0 * t, 1.0 + 0.5 * np.sin(6 * t)], 1)
    cloud = (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)
    task = dict(adj_thresh=0.5, topk=10, connect_tools_all=False, sim_real_ratio=10, push_length=0.1, gripper_enable=False,
                max_n=1, max_nR=8000, n_his=4, eef_num=1, material="rope", pusher_points=[[0.0, 0.0, 0.12]])
    return cloud, task


def _shipped_rope_samples(B, rng):
    lo, hi = np.float32([-4.5, -2.5, -3.14, 5.0]), np.float32([0.0, 4.5, 3.14, 15.0])     # planning/rope.yaml:28-29
    return (rng.random((B, 1, 4), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)     # plan_utils.py:48-50


def test_oracle_rollout_work_agrees_with_the_oracles_own_edge_lists():
    from oracle import adaptigraph_oracle as O
    rng = np.random.default_rng(11)
    W = O.random_weights(11)
    cloud, task = _rope_case(60, rng)
    a = _shipped_rope_samples(40, rng)
    a[:, 0, 3] = rng.integers(3, 8, 40) + 0.5
    a[:8, 0, 0] = cloud[30, 0] + rng.uniform(-0.9, 0.9, 8)                  # some start near the rope: contact after a few forwards
    a[:8, 0, 1] = cloud[30, 2] + rng.uniform(-0.9, 0.9, 8)
    work, first = O.rollout_work(W, 3, cloud, a, task)
    tr = []
    O.dynamics(W, 3, cloud, a, task, trace=tr)
    N_o = cloud.shape[0]
    for b in range(40):
        hit = next((f + 1 for f, rec in enumerate(tr[b]) if ((rec["recv"] >= N_o) | (rec["send"] >= N_o)).any()), 0)
        assert first[b] == hit, (b, first[b], hit)
        rep = int(a[b, 0, 3])
        assert work[b] == (rep - hit + 1 if hit else 0)
    assert (first == 0).any() and (first > 1).any() and (first == 1).any(), first


def test_work_balanced_shards_on_the_shipped_rope_sampler():
    """2000 pushes drawn as the reference's planner draws them over the rope task's action box, rope of 200 particles: most
    never touch.  Work-balanced shards stay within 10 % of each other in work; count-balanced ones do not."""
    from oracle import adaptigraph_oracle as O
    from adaptigraph_amd.sharding import work_balanced_bounds, WORK_FLOOR
    rng = np.random.default_rng(12)
    W = O.random_weights(12)
    cloud, task = _rope_case(200, rng)
    a = _shipped_rope_samples(2000, rng)
    work, first = O.rollout_work(W, 3, cloud, a, task)
    free = float((first == 0).mean())
    assert 0.6 < free < 0.95, free                                           # (79 % on the bench planner's cloud)
    cost = work + WORK_FLOOR
    for world in (2, 4, 8):
        bounds = work_balanced_bounds(work, world)
        assert bounds[0][0] == 0 and bounds[-1][1] == 2000 and all(x[1] == y[0] for x, y in zip(bounds, bounds[1:]))
        loads = np.array([cost[lo:hi].sum() for lo, hi in bounds])
        even = np.array([cost[lo:hi].sum() for lo, hi in (shard_bounds(2000, world, r) for r in range(world))])
        print(f"world {world}: work-balanced max/mean {loads.max() / loads.mean():.3f}, count-balanced {even.max() / even.mean():.3f}, "
              f"{free:.0%} of the pushes never touch")
        assert loads.max() / loads.mean() <= 1.10 and loads.min() / loads.mean() >= 0.90, (world, loads)
    assert even.max() / even.mean() > 1.10                                   # (at 8 ranks the count-balanced split is visibly worse)


def _work_worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import adaptigraph_oracle as O
    from oracle import costs_oracle as Cc
    from adaptigraph_amd.sharding import sharded_candidate_rewards, work_balanced_bounds
    from adaptigraph_amd.losses import _global_max
    rng = np.random.default_rng(13)                       # every rank builds the same inputs
    W = O.random_weights(13)
    cloud, task = _rope_case(48, rng)
    target = (cloud + np.float32([0.3, 0, 0.2])).astype(np.float32)
    acts = _shipped_rope_samples(B, rng)
    acts[:, 0, 3] = rng.integers(2, 5, B) + 0.5
    acts[: B // 4, 0, 0] = cloud[24, 0] + rng.uniform(-0.6, 0.6, B // 4)   # a heavy head: the first quarter touches
    acts[: B // 4, 0, 1] = cloud[24, 2] + rng.uniform(-0.6, 0.6, B // 4)
    actions = torch.from_numpy(acts)
    calls = []

    def rollout(a):                                       # the oracle stands in for the engine (stepping every forward: same values)
        calls.append(int(a.shape[0]))
        return torch.from_numpy(O.dynamics(W, 3, cloud, a.numpy(), task)["state_seqs"])

    def work_fn(a):
        return O.rollout_work(W, 3, cloud, a.numpy(), task)[0]

    def make_reward(group):
        def reward(seq, a):
            s = seq.numpy()
            b, H = s.shape[:2]
            err = torch.from_numpy(Cc.chamfer(s.reshape(b * H, -1, 3), target[None]).reshape(b, H).astype(np.float32))
            w = (2.0 / (_global_max(err, group).to(torch.float64) + 1e-6)).to(torch.float32)          # plan.py:37
            return -w * err[:, -1]
        return reward

    full = sharded_candidate_rewards(actions, rollout, make_reward(True), work_fn=work_fn)
    want = make_reward(None)(rollout(actions), actions)
    bounds = work_balanced_bounds(work_fn(actions), world)
    ok = torch.equal(full, want) and calls[0] == bounds[rank][1] - bounds[rank][0]
    uneven = len({hi - lo for lo, hi in bounds}) > 1       # really cut by work: the shards hold different numbers of candidates
    q.put((rank, bool(ok), bool(uneven)))
    dist.destroy_process_group()


def test_gloo_work_balanced_rewards_match_unsharded():
    ctx = mp.get_context("spawn")
    for world in (2, 4):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_work_worker, args=(r, world, port, 24, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res == [(r, True, True) for r in range(world)], res


# ------------------------------------------------------------------------------------- r06: one rank's bounds, mixed batches
def test_work_balanced_bounds_leave_no_rank_without_a_candidate():
    from adaptigraph_amd.sharding import work_balanced_bounds
    assert work_balanced_bounds(np.r_[1000.0, np.zeros(7)], 4) == [(0, 1), (1, 2), (2, 3), (3, 8)]      # one heavy head
    assert work_balanced_bounds(np.r_[np.zeros(7), 1000.0], 4) == [(0, 5), (5, 6), (6, 7), (7, 8)]      # one heavy tail
    for world in (2, 3, 8):
        for n in (world, world + 1, 50):
            b = work_balanced_bounds(np.random.default_rng(n).integers(0, 9, n), world)
            assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(hi > lo for lo, hi in b)


def test_mixed_shard_bounds_cut_all_materials_at_once():
    """BASELINE configs[4]: 172 rope + 170 granular + 170 cloth candidates with their own particle counts; a candidate's work
    estimate is N_b x (topk + M) of its material.  One cut over the concatenated batch: every candidate lands on exactly one rank,
    ranks carry about the same work, and a rank's piece may span materials."""
    from adaptigraph_amd.sharding import mixed_shard_bounds
    rng = np.random.default_rng(4)
    spec = (("rope", 172, 300, 10 + 1), ("granular", 170, 1024, 20 + 5), ("cloth", 170, 2025, 5 + 1))
    w = [rng.integers(N // 2, N + 1, B).astype(np.float64) * k for _, B, N, k in spec]
    total = sum(x.sum() for x in w)
    for world in (1, 2, 4, 8):
        table = mixed_shard_bounds(w, world)
        assert len(table) == world and all(len(row) == 3 for row in table)
        for m, (_, B, _, _) in enumerate(spec):             # every batch is covered once, in rank order
            spans = [row[m] for row in table]
            assert spans[0][0] == 0 and spans[-1][1] == B and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        loads = np.array([sum(w[m][lo:hi].sum() for m, (lo, hi) in enumerate(row)) for row in table])
        assert abs(loads.sum() - total) < 1e-6 and loads.max() - loads.min() <= 2 * max(x.max() for x in w) + 1e-6, (world, loads)
    two = mixed_shard_bounds(w, 2)
    assert two[0][0] == (0, 172) and two[0][1][1] > 0 and two[1][0] == (172, 172)      # rank 0: all the rope and some granular
    # count-balanced per material (what sharding each material separately by count would give) is visibly worse
    per_count = np.array([sum(w[m][lo:hi].sum() for m, (lo, hi) in enumerate([shard_bounds(B, 8, r) for _, B, _, _ in spec])) for r in range(8)])
    eight = mixed_shard_bounds(w, 8)
    loads8 = np.array([sum(w[m][lo:hi].sum() for m, (lo, hi) in enumerate(row)) for row in eight])
    assert loads8.max() / loads8.mean() <= per_count.max() / per_count.mean() + 1e-9


def _mixed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptigraph_amd.sharding import sharded_mixed_values, sharded_candidate_rewards, mixed_shard_bounds
    rng = np.random.default_rng(6)                         # every rank builds the same inputs
    sizes, scale = (23, 17, 19), (11.0, 25.0, 6.0)
    counts = [rng.integers(5, 40, n) for n in sizes]
    w = [c.astype(np.float64) * k for c, k in zip(counts, scale)]
    states = [torch.from_numpy(rng.normal(size=(n, 40, 3)).astype(np.float32)) for n in sizes]
    seen = []

    def evaluate_rows(m, lo, hi):                          # stands in for dynamics_masked + a per-candidate cost: row-wise
        seen.append((m, lo, hi))
        return (states[m][lo:hi] * (m + 1)).sum((1, 2))

    got = sharded_mixed_values(w, evaluate_rows)
    want = [(states[m] * (m + 1)).sum((1, 2)) for m in range(3)]
    ok = all(torch.equal(g_, w_) for g_, w_ in zip(got, want))
    ok = ok and seen == [(m, lo, hi) for m, (lo, hi) in enumerate(mixed_shard_bounds(w, world)[rank]) if hi > lo]
    # a work estimate that differs between the ranks (per-context state) must not desynchronise the all-gather: rank 0's cuts serve
    torch.manual_seed(0)
    actions = torch.rand(31, 2, 4)
    work_fn = lambda a: np.arange(31) % (3 + rank)         # a different estimate on every rank
    full = sharded_candidate_rewards(actions, lambda a: a.sum(-1, keepdim=True).repeat(1, 1, 5), lambda seq, a: seq[:, -1].mean(-1),
                                     work_fn=work_fn)
    ok = ok and torch.equal(full, actions.sum(-1, keepdim=True).repeat(1, 1, 5)[:, -1].mean(-1))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gloo_mixed_batch_values_and_rank0_bounds():
    ctx = mp.get_context("spawn")
    for world in (2, 4, 8):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res == [(r, True) for r in range(world)], res
