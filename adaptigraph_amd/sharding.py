"""Candidate sharding across the GPUs of one node.

MPC candidates are independent, so rank r rolls out the contiguous shard [lo, hi) of the batch and the only exchange
is one all-gather of the per-candidate costs (B/G fp32 per rank: 512 B at B=1024, G=8) so that every rank can finish
the MPPI update identically.  Backend-agnostic: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
The reference has no multi-GPU code at all; this is new, not a port.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_candidates: int, world: int, rank: int):
    """Contiguous, balanced shards; the first (n % world) ranks get one extra candidate."""
    base, extra = divmod(n_candidates, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_costs(cost_local: torch.Tensor, n_candidates: int, group=None) -> torch.Tensor:
    """cost_local: (hi-lo,) costs of this rank's shard -> (n_candidates,) costs of the whole batch on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert cost_local.numel() == n_candidates
        return cost_local
    world = dist.get_world_size(group)
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


def sharded_rollout_costs(rollout_fn, cost_fn, actions: torch.Tensor, group=None):
    """Run `rollout_fn(actions[lo:hi])` on this rank's shard, reduce it to per-candidate costs with `cost_fn`, and
    return the full (B,) cost vector on every rank.  `actions` is the FULL (B, H, 4) batch, identical on all ranks."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(actions.shape[0], world, rank)
    local = cost_fn(rollout_fn(actions[lo:hi]))
    return all_gather_costs(local, actions.shape[0], group)
