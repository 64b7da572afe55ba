"""Candidate sharding across the GPUs of one node.

MPC candidates are independent, so rank r rolls out the contiguous shard [lo, hi) of the batch and the only exchange
is one all-gather of the per-candidate costs (B/G fp32 per rank: 512 B at B=1024, G=8) so that every rank can finish
the MPPI update identically.  Backend-agnostic: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
The reference has no multi-GPU code at all; this is new, not a port.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

# True: a world of ONE rank still issues its all-gather (bench.py AG_BENCH_FORCE_DIST=1: executes the RCCL calls on a one-GPU box)
FORCE_COLLECTIVES = False


def shard_bounds(n_candidates: int, world: int, rank: int):
    """Contiguous, balanced shards; the first (n % world) ranks get one extra candidate."""
    base, extra = divmod(n_candidates, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_bounds_weighted(weights, world: int, rank: int):
    """Contiguous shards of a RAGGED batch balanced by work instead of by count (for mixed-size graphs
    the work of a candidate is proportional to its edge count, or to N_b * (topk + M) before the graph exists).
    Cut r lies where the running sum first reaches r/world of the total (nearest candidate boundary); every rank
    computes the same cuts from the same `weights`, so no exchange is needed."""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    assert w.ndim == 1 and (w >= 0).all()
    n = w.shape[0]
    csum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        target = csum[-1] * r / world
        i = int(np.searchsorted(csum, target))                      # first boundary with csum >= target
        if i > 0 and target - csum[i - 1] <= csum[min(i, n)] - target:
            i -= 1                                                  # the boundary before it is closer
        cuts.append(min(n, max(cuts[-1], i)))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


def mixed_shard_bounds(weights_per_batch, world: int):
    """Shards of a MIXED batch (BASELINE configs[4]: several materials, each a ragged batch of its own) cut by work over ALL
    materials at once: the candidates are taken in batch order (all of batch 0, then batch 1, ...), the weights concatenated -
    a candidate's work estimate is N_b * (topk + M) of ITS material, so a cloth candidate weighs ~6x a rope one - and the
    concatenated sequence is cut into `world` contiguous pieces by shard_bounds_weighted.  Returns [rank][batch] -> (lo, hi):
    the rows of every batch a rank evaluates (empty ranges where a rank's piece does not reach into a batch).  Every rank
    computes the same table from the same numbers."""
    import numpy as np
    sizes = [len(w) for w in weights_per_batch]
    flat = np.concatenate([np.asarray(w, dtype=np.float64) for w in weights_per_batch]) if sizes else np.zeros(0)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    table = []
    for r in range(world):
        lo, hi = shard_bounds_weighted(flat, world, r)
        table.append([(int(min(max(lo, a), b) - a), int(min(max(hi, a), b) - a)) for a, b in zip(starts[:-1], starts[1:])])
    return table


def sharded_mixed_values(weights_per_batch, evaluate_rows, group=None):
    """Per-candidate values of a mixed batch with every rank evaluating its mixed_shard_bounds piece: evaluate_rows(m, lo, hi) ->
    (hi - lo,) tensor for rows [lo, hi) of batch m (called only for non-empty ranges).  -> [full (B_m,) tensor per batch] on
    every rank: one all-gather of the rank's concatenated values, variable-length pieces."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    table = mixed_shard_bounds(weights_per_batch, world)
    sizes = [len(w) for w in weights_per_batch]
    pieces = [evaluate_rows(m, lo, hi) for m, (lo, hi) in enumerate(table[rank]) if hi > lo]
    if pieces:
        local = torch.cat([p.reshape(-1) for p in pieces])
    else:                                                   # a rank without rows still takes part in the all-gather
        local = torch.zeros(0, device="cuda" if distributed and dist.get_backend(group) == "nccl" else "cpu")
    starts = [0]
    for n in sizes:
        starts.append(starts[-1] + n)
    bounds = []
    for r in range(world):
        rows = [(starts[m] + lo, starts[m] + hi) for m, (lo, hi) in enumerate(table[r]) if hi > lo]
        bounds.append((rows[0][0], rows[-1][1]) if rows else (bounds[-1][1] if bounds else 0,) * 2)
    full = all_gather_costs(local.contiguous(), starts[-1], group, bounds=bounds) if world > 1 else local
    return [full[starts[m]:starts[m + 1]] for m in range(len(sizes))]


def all_gather_costs(cost_local: torch.Tensor, n_candidates: int, group=None, bounds=None) -> torch.Tensor:
    """cost_local: (hi-lo,) costs of this rank's shard -> (n_candidates,) costs of the whole batch on every rank.
    bounds: optional list of (lo, hi) per rank for work-balanced shards (default: shard_bounds)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        assert cost_local.numel() == n_candidates
        return cost_local
    world = dist.get_world_size(group)
    if bounds is not None:
        per = max(hi - lo for lo, hi in bounds)
        padded = torch.zeros(per, device=cost_local.device, dtype=cost_local.dtype)
        padded[:cost_local.numel()] = cost_local
        out = torch.empty(per * world, device=cost_local.device, dtype=cost_local.dtype)
        dist.all_gather_into_tensor(out, padded, group=group)
        return torch.cat([out[r * per: r * per + (hi - lo)] for r, (lo, hi) in enumerate(bounds)])
    per = (n_candidates + world - 1) // world
    padded = torch.zeros(per, device=cost_local.device, dtype=cost_local.dtype)
    padded[:cost_local.numel()] = cost_local
    out = torch.empty(per * world, device=cost_local.device, dtype=cost_local.dtype)
    dist.all_gather_into_tensor(out, padded, group=group)
    if n_candidates % world == 0:
        return out
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(n_candidates, world, r)
        pieces.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(pieces)


def sharded_rollout_costs(rollout_fn, cost_fn, actions: torch.Tensor, group=None, weights=None):
    """Run `rollout_fn(actions[lo:hi])` on this rank's shard, reduce it to per-candidate costs with `cost_fn`, and
    return the full (B,) cost vector on every rank.  `actions` is the FULL (B, H, 4) batch, identical on all ranks.
    weights: optional per-candidate work estimate (ragged batches) -> work-balanced shards; the callbacks then take
    (lo, hi) as extra arguments since per-candidate inputs other than the actions must be sliced too."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if weights is None:
        lo, hi = shard_bounds(actions.shape[0], world, rank)
        local = cost_fn(rollout_fn(actions[lo:hi]))
        return all_gather_costs(local, actions.shape[0], group)
    bounds = [shard_bounds_weighted(weights, world, r) for r in range(world)]
    lo, hi = bounds[rank]
    local = cost_fn(rollout_fn(actions[lo:hi], lo, hi), lo, hi)
    return all_gather_costs(local, actions.shape[0], group, bounds=bounds)


# what a candidate costs beside its model forwards (plan kernels, slot init / the copy of a base state), in forwards: keeps a
# shard of candidates that are never stepped from growing without bound
WORK_FLOOR = 0.25


def work_balanced_bounds(work, world: int):
    """[(lo, hi)] per rank: contiguous shards of the candidate batch cut by WORK (model forwards per candidate, e.g.
    forward_dynamics.rollout_work) instead of by count.  With the contact-free prefix a candidate's work is 0 .. action_repeat
    forwards depending on its first contact (the shipped rope sampler: 79 % of the pushes never touch), so count-balanced shards
    can differ several-fold in work.  Every rank computes the same cuts from the same numbers."""
    import numpy as np
    w = np.asarray(work, dtype=np.float64) + WORK_FLOOR
    cuts = [shard_bounds_weighted(w, world, r)[0] for r in range(world)] + [len(w)]
    if len(w) >= world:                                    # no rank is left without a candidate (a rollout of zero rows)
        for r in range(1, world):
            cuts[r] = max(cuts[r], cuts[r - 1] + 1)
        for r in range(world - 1, 0, -1):
            cuts[r] = min(cuts[r], cuts[r + 1] - 1)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def bounds_of_rank0(bounds, group=None):
    """[(lo, hi)] per rank as RANK 0 computed them, on every rank (one broadcast of world + 1 integers).  Work-balanced bounds come
    out of a per-rank work estimate - forward_dynamics.rollout_work goes through the rank's own engine context, whose kept base
    rollout and census verdict decide whether the estimate counts prefix-shared forwards - so two ranks can arrive at different
    cuts from the same batch; bounds that size an all-gather must be ONE rank's (different layouts would permute the rewards
    silently, different sizes would hang).  Any contiguous cut gives the same rewards bit for bit; only the balance is rank 0's."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bounds
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([b[0] for b in bounds] + [bounds[-1][1]], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    cuts = t.tolist()
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(len(bounds))]


def sharded_candidate_rewards(actions: torch.Tensor, rollout_fn, reward_fn, group=None, work_fn=None) -> torch.Tensor:
    """One MPC evaluation of a candidate batch sharded over the ranks of `group` - the step bench.py times.

    actions (B, H, 4): the FULL batch, identical on every rank (same seed).  rollout_fn(actions_local) -> state_seqs of
    the local shard; reward_fn(state_seqs, actions_local) -> (b,) rewards (any batch-global quantity inside it - the
    error maximum of running_cost, the cloth penalty's distance maximum - must be all-reduced by the callee, cf.
    losses._global_max).  Returns the (B,) reward vector of the whole batch on every rank: contiguous shards, one
    all-gather of B/G floats per rank, nothing else crosses ranks.
    work_fn(actions) -> (B,) work per candidate, identical on every rank (forward_dynamics.rollout_work bound to the model):
    shards are then cut by work (work_balanced_bounds) instead of by count; None / one rank: count-balanced."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    bounds = None
    if actions.shape[0] < world:
        raise ValueError(f"{actions.shape[0]} candidates cannot be sharded over {world} ranks (a rank would roll out nothing)")
    if work_fn is not None and world > 1:
        bounds = work_balanced_bounds(work_fn(actions), world)
        bounds = bounds_of_rank0(bounds, group)              # work_fn runs per rank: one rank's cuts size the all-gather
        lo, hi = bounds[rank]
    else:
        lo, hi = shard_bounds(actions.shape[0], world, rank)
    local = actions[lo:hi]
    rewards = reward_fn(rollout_fn(local), local)
    return all_gather_costs(rewards.contiguous(), actions.shape[0], group, bounds=bounds)
