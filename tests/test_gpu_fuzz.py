"""Seeded random configurations of the rollout engine against the oracle (-m gpu): particle counts from 1 to a few hundred,
1-point and 5-point pushers, batches of 1..40 candidates, 1..3 look-ahead steps, repeats 0..4, top-k below / above the
particle count, masked and unmasked variants - each run (a) through the default engine settings against the numpy oracle and
(b) again under a random mix of the bit-identical execution paths (streams, chunk size, latency chains, launch order,
CSR / slot-indexed graphs, device-planned actions), which must reproduce (a) bit for bit (device-planned: the states to 1e-5,
its own two launch orders bit for bit)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _ppm, POS_TOL
from test_gpu_more import _task, _grid, _rope, _actions, _model

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    material = ["rope", "granular", "cloth"][seed % 3]
    if material == "rope":
        n = int(rng.integers(1, 260))
        cloud = _rope(max(n, 2), rng)[:n]
    else:
        side = int(rng.integers(1, 17))
        cloud = _grid(side, 0.12 if material == "granular" else 0.3, 0.02, rng)
        cloud = cloud[:int(rng.integers(1, side * side + 1))]
    over = {}
    if seed % 5 == 0:
        over["topk"] = int(rng.integers(1, 4))                          # tight top-k
    if seed % 7 == 0:
        over["topk"] = 1000                                             # top-k above the particle count: radius only (CSR path)
    if seed % 4 == 1:
        over["connect_tools_all"] = not _task(material)["connect_tools_all"]
    task = _task(material, max_nR=60000, **over)
    B, H = int(rng.integers(1, 41)), int(rng.integers(1, 4))
    reps = rng.integers(0, 5, (B, H))
    a = _actions(cloud, B, H, np.maximum(reps, 0), rng, spread=0.5)
    a[..., 3] = reps + rng.uniform(0.05, 0.95, reps.shape)              # int(length) = repeat, lengths not on the .5 grid
    return rng, material, task, cloud, a.astype(np.float32), reps


@pytest.mark.parametrize("seed", range(18))
def test_random_rollout_configurations(seed):
    import adaptigraph_amd as ag
    from oracle import adaptigraph_oracle as O
    dev = torch.device("cuda:0")
    rng, material, task, cloud, a_np, reps = _case(seed)
    W, m = _model(ag, O, material, 200 + seed, dev)
    eng = m.engine(dev)
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, material)
    base = ag.dynamics(s0, a, m, dev, ppm)
    assert eng.rollout_counts() == (int(reps.sum()), int(reps.sum()))
    want = O.dynamics(W, 3, cloud, a_np, task)
    assert np.array_equal(base["action_seqs"].cpu().numpy(), want["action_seqs"])
    err = np.abs(base["state_seqs"].cpu().numpy() - want["state_seqs"]).reshape(len(a_np), -1).max(1)
    assert (err <= POS_TOL).mean() >= 0.9, (seed, material, cloud.shape, err)
    opts = dict(streams=int(rng.integers(1, 5)), latency=int(rng.integers(-1, 2)), repeat_sort=int(rng.integers(0, 2)),
                ell_graph=int(rng.integers(0, 2)), self_dedupe=int(rng.integers(0, 2)), edge_block_min=int(rng.choice([-1, 1, 10 ** 9])),
                zigzag=int(rng.integers(0, 2)), share_first=int(rng.integers(0, 2)), share_prefix=int(rng.integers(0, 2)))
    eng.set_chunk(int(rng.integers(0, len(a_np) + 1)))
    try:
        with eng.options(**opts):
            again = ag.dynamics(s0, a, m, dev, ppm)
            assert torch.equal(again["state_seqs"], base["state_seqs"]), (seed, opts)
            tdev = dict(task, action_upper_lim=[0.0, 4.5, 3.14, 4.0])
            d1 = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
            ex, need = eng.rollout_counts()
            assert need == int(reps.sum()) and (ex == need or not opts["repeat_sort"] or (opts["share_prefix"] and ex < need + 5))
            with eng.options(repeat_sort=1 - opts["repeat_sort"], streams=1):
                d2 = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
            assert torch.equal(d1["state_seqs"], d2["state_seqs"]) and torch.equal(d1["action_seqs"], d2["action_seqs"])
    finally:
        eng.set_chunk(0)
    assert float((d1["action_seqs"] - base["action_seqs"]).abs().max()) <= 1e-6
    e2 = (d1["state_seqs"] - base["state_seqs"]).abs().reshape(len(a_np), -1).max(1).values
    assert float((e2 <= POS_TOL).float().mean()) >= 0.9, (seed, e2)


@pytest.mark.parametrize("seed", range(8))
def test_random_masked_configurations(seed):
    import adaptigraph_amd as ag
    from oracle import adaptigraph_oracle as O
    dev = torch.device("cuda:0")
    rng, material, task, cloud, a_np, reps = _case(100 + seed)
    n = cloud.shape[0]
    B = len(a_np)
    state = np.zeros((B, n, 3), np.float32)
    mask = np.zeros((B, n), bool)
    for b in range(B):
        c = int(rng.integers(1, n + 1))
        keep = np.sort(rng.choice(n, c, replace=False))
        state[b, :c] = cloud[keep]
        state[b, c:] = rng.normal(0, 1, (n - c, 3))                     # padding rows need not be zero
        mask[b, :c] = True
        if seed % 2 and c > 2:
            mask[b, int(rng.integers(0, c))] = False                     # a hole inside the prefix
    act = a_np[:, 0]
    W, m = _model(ag, O, material, 300 + seed, dev)
    eng = m.engine(dev)
    args = (torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(act).to(dev))
    ppm = _ppm(task, material)
    base = ag.dynamics_masked(*args, m, dev, ppm)["state_seqs"]
    assert eng.rollout_counts() == (int(reps[:, 0].sum()), int(reps[:, 0].sum()))      # masked batches too: no surplus forward
    want = O.dynamics_masked(W, 3, state, mask, act, task)["state_seqs"]
    err = np.abs(base.cpu().numpy() - want).reshape(B, -1).max(1)
    assert (err <= POS_TOL).mean() >= 0.9, (seed, material, err)
    eng.set_chunk(int(rng.integers(0, B + 1)))
    try:
        with eng.options(ragged=int(rng.integers(0, 2)), streams=int(rng.integers(1, 3)), latency=int(rng.integers(-1, 2)),
                         ell_graph=int(rng.integers(0, 2)), repeat_sort=int(rng.integers(0, 2))):
            again = ag.dynamics_masked(*args, m, dev, ppm)["state_seqs"]
    finally:
        eng.set_chunk(0)
    assert torch.equal(again, base), seed
