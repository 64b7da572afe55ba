"""running_cost with the reference's signature (src/planning/plan.py:27-59): the evaluate_traj_fn the planner binds.

The particle reductions (error function, penalty function, x/z bounds) run in HIP kernels; what is left is arithmetic
on (B, H) tensors.  `group`: when the candidate batch is sharded over ranks, the batch-global maximum of the error
(plan.py:37) is all-reduced so every rank weights its candidates exactly like the unsharded call.
"""
from __future__ import annotations

import ctypes as C

import torch

from .context import default_engine, ptr, current_stream
from .losses import state_stats, _global_max


@torch.no_grad()
def running_cost(state, action, state_cur, error_func, penalty_func, bbox, group=None, **kwargs):
    """state (B,H,N,3), action (B,H,4) raw, state_cur (N,3) -> {'reward_seqs': (B,)}"""
    bsz, n_look_forward = state.shape[0], state.shape[1]
    state_flat = state.reshape(bsz * n_look_forward, state.shape[2], state.shape[3])
    dev = state.device
    # (.to(dev): raw pointers go to a HIP kernel below - a callable that returns a CPU tensor must not hand it a host address)
    error = error_func(state_flat).reshape(bsz, n_look_forward).to(dev, torch.float32).contiguous()      # :35-36
    collision_penalty = penalty_func(state, action, state_cur).to(dev, torch.float32).contiguous()       # :39
    assert collision_penalty.shape == (bsz, n_look_forward)
    st = state_stats(state_flat)                                                       # :41-44 in one pass: (B*H, 5)
    # :37, :45-53 in ONE launch (csrc/ag_cost.hip: k_reward): error_weight = 2.0 / (error.max() + 1e-6) formed in double
    # precision and rounded to fp32 only when it scales the error, as the reference's Python float does; the box penalty; the
    # two means; the reward.  A sharded batch all-reduces the error maximum first and hands it in.
    emax = None if group is None else _global_max(error, group).reshape(1).to(dev, torch.float32).contiguous()
    bb = torch.as_tensor(bbox).to("cpu", torch.float64)
    bbox4 = (C.c_double * 4)(float(bb[0, 0]), float(bb[0, 1]), float(bb[1, 0]), float(bb[1, 1]))
    reward = torch.empty(bsz, device=dev, dtype=torch.float32)
    eng = default_engine(dev)
    eng.check(eng.lib.ag_cost_reward(eng.ctx, current_stream(dev), ptr(error), ptr(collision_penalty), ptr(st), ptr(emax), bbox4,
                                     bsz, n_look_forward, ptr(reward)))
    return {"reward_seqs": reward}
