"""running_cost with the reference's signature (src/planning/plan.py:27-59): the evaluate_traj_fn the planner binds.

The particle reductions (error function, penalty function, x/z bounds) run in HIP kernels; what is left is arithmetic
on (B, H) tensors.  `group`: when the candidate batch is sharded over ranks, the batch-global maximum of the error
(plan.py:37) is all-reduced so every rank weights its candidates exactly like the unsharded call.
"""
from __future__ import annotations

import torch

from .losses import state_stats, _global_max


@torch.no_grad()
def running_cost(state, action, state_cur, error_func, penalty_func, bbox, group=None, **kwargs):
    """state (B,H,N,3), action (B,H,4) raw, state_cur (N,3) -> {'reward_seqs': (B,)}"""
    bsz, n_look_forward = state.shape[0], state.shape[1]
    state_flat = state.reshape(bsz * n_look_forward, state.shape[2], state.shape[3])
    error = error_func(state_flat).reshape(bsz, n_look_forward)                        # :35-36
    # :37 forms 2.0 / (error.max().item() + 1e-6) in Python double precision and rounds to fp32 only when it scales the
    # error; the same arithmetic on a float64 device scalar needs no host sync
    error_weight = (2.0 / (_global_max(error, group).to(torch.float64) + 1e-6)).to(torch.float32)
    collision_penalty = penalty_func(state, action, state_cur)                         # :39
    st = state_stats(state_flat).reshape(bsz, n_look_forward, 5)                       # :41-44 in one pass
    xmin, xmax, zmin, zmax = st[..., 1], st[..., 2], st[..., 3], st[..., 4]
    bb = torch.as_tensor(bbox).to("cpu", torch.float64)
    zero = torch.zeros_like(xmin)
    box_penalty = torch.stack([torch.maximum(xmin - float(bb[0, 0]), zero), torch.maximum(float(bb[0, 1]) - xmax, zero),
                               torch.maximum(zmin - float(bb[1, 0]), zero), torch.maximum(float(bb[1, 1]) - zmax, zero)],
                              dim=-1)                                                  # :45-50
    box_penalty = torch.exp(-box_penalty * 100.0).max(dim=-1).values                   # :51
    reward = -error_weight * error[:, -1] - 5.0 * collision_penalty.mean(dim=1) - 5.0 * box_penalty.mean(dim=1)   # :53
    return {"reward_seqs": reward}
