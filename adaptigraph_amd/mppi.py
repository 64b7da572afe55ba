"""MPPI sampling / update with the reference's names and arithmetic (src/planning/plan_utils.py:31-101) and one whole
MPC iteration (src/planning/real_world/planner.py:234-277) wired to the HIP engine.

The sampling and the softmax-weighted update are O(B*H*4) elementwise work on torch tensors; the op order is the
reference's, so with the same torch generator state on the same device the results are bit-identical to it.
What the engine changes is upstream: the reference evaluates n_sample = 20000 in 40 host-side chunks of 500
(plan.py:177-182, 241-247) because its dense rollout does not fit; here one call takes the whole batch (the engine
chunks on the device), optionally sharded over the ranks of a torch.distributed group.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def angle_normalize(x):
    return ((x + math.pi) % (2 * math.pi)) - math.pi                                  # plan_utils.py:31-32


def clip_actions(action, action_lower_lim, action_upper_lim):
    action_new = action.clone()                                                        # plan_utils.py:35-39
    action_new[..., 2] = angle_normalize(action[..., 2])
    action_new.data.clamp_(action_lower_lim, action_upper_lim)
    return action_new


def sample_action_seq(act_seq, action_lower_lim, action_upper_lim, n_sample, device, iter_index=0, noise_level=0.3,
                      push_length=0.10):
    """plan_utils.py:42-77 -> (n_sample, n_look_ahead, 4)"""
    if iter_index == 0:                                                                # resample completely
        return torch.rand((n_sample, act_seq.shape[0], act_seq.shape[1]), device=device) * \
            (action_upper_lim - action_lower_lim) + action_lower_lim
    n_look_ahead = act_seq.shape[0]
    assert act_seq.shape[-1] == 4
    act_seqs = torch.stack([act_seq.clone()] * n_sample)
    xs, ys, thetas, lengths = act_seqs[:, :, 0], act_seqs[:, :, 1], act_seqs[:, :, 2], act_seqs[:, :, 3]
    x_ends = xs - lengths * push_length * torch.cos(thetas)
    y_ends = ys - lengths * push_length * torch.sin(thetas)
    for i in range(n_look_ahead):
        noise_sample = torch.normal(0, noise_level, (n_sample, 4), device=device)
        act_residual = (0.1 * (10 ** i)) * noise_sample
        xs_i = xs[:, i] + act_residual[:, 0]
        ys_i = ys[:, i] + act_residual[:, 1]
        x_ends_i = x_ends[:, i] + act_residual[:, 2]
        y_ends_i = y_ends[:, i] + act_residual[:, 3]
        thetas_i = torch.atan2(ys_i - y_ends_i, xs_i - x_ends_i)
        lengths_i = torch.norm(torch.stack([x_ends_i - xs_i, y_ends_i - ys_i], dim=-1), dim=-1).clone() / push_length
        act_seq_i = clip_actions(torch.stack([xs_i, ys_i, thetas_i, lengths_i], dim=-1), action_lower_lim, action_upper_lim)
        act_seqs[1:, i] = act_seq_i[1:].clone()                                        # sample 0 keeps the nominal action
    return act_seqs


def optimize_action_mppi(act_seqs, reward_seqs, reward_weight=100.0, action_lower_lim=None, action_upper_lim=None,
                         push_length=0.10):
    """plan_utils.py:80-101: softmax-weighted average of start and end points, re-encoded as (x, y, theta, length)."""
    weight_seqs = F.softmax(reward_seqs * reward_weight, dim=0).unsqueeze(-1)
    assert act_seqs.shape[-1] == 4
    xs, ys, thetas, lengths = act_seqs[:, :, 0], act_seqs[:, :, 1], act_seqs[:, :, 2], act_seqs[:, :, 3]
    x_ends = xs - lengths * push_length * torch.cos(thetas)
    y_ends = ys - lengths * push_length * torch.sin(thetas)
    x = torch.sum(weight_seqs * xs, dim=0)
    y = torch.sum(weight_seqs * ys, dim=0)
    x_end = torch.sum(weight_seqs * x_ends, dim=0)
    y_end = torch.sum(weight_seqs * y_ends, dim=0)
    theta = torch.atan2(y - y_end, x - x_end)
    length = torch.norm(torch.stack([x_end - x, y_end - y], dim=-1), dim=-1) / push_length
    return clip_actions(torch.stack([x, y, theta, length], dim=-1), action_lower_lim, action_upper_lim)


@torch.no_grad()
def mpc_iteration(state_cur, act_seq, model_rollout_fn, evaluate_traj_fn, action_lower_lim, action_upper_lim, n_sample,
                  device, noise_level=1.0, reward_weight=500.0, push_length=0.10, iter_index=0, rollout_best=True,
                  act_seqs=None, group=None):
    """Planner.trajectory_optimization_mppi with n_update_iter = 1 (planner.py:234-277, plan.py:199): sample ->
    rollout -> evaluate -> MPPI update -> best candidate -> (optionally) roll the best out again.

    model_rollout_fn(state_cur, act_seqs) and evaluate_traj_fn(state_seqs, act_seqs, state_cur=...) are the same
    partials plan.py builds (plan.py:175, 190).  `group`: shard the candidates over the ranks of that group; every
    rank must call with the same generator state (or pass the same `act_seqs`)."""
    from .sharding import shard_bounds, all_gather_costs
    import torch.distributed as dist
    if act_seqs is None:
        act_seqs = sample_action_seq(act_seq, action_lower_lim, action_upper_lim, n_sample, device,
                                     iter_index=iter_index, noise_level=noise_level, push_length=push_length)
    world = dist.get_world_size(None if group is True else group) if group is not None else 1
    rank = dist.get_rank(None if group is True else group) if group is not None else 0
    lo, hi = shard_bounds(act_seqs.shape[0], world, rank)
    model_out = model_rollout_fn(state_cur, act_seqs[lo:hi])
    eval_out = evaluate_traj_fn(model_out["state_seqs"], act_seqs[lo:hi], state_cur=state_cur)
    reward_seqs = all_gather_costs(eval_out["reward_seqs"].contiguous(), act_seqs.shape[0],
                                   None if group in (None, True) else group) if world > 1 else eval_out["reward_seqs"]
    new_act_seq = optimize_action_mppi(act_seqs, reward_seqs, reward_weight, action_lower_lim, action_upper_lim, push_length)
    best = torch.argmax(reward_seqs)
    best_act_seq, best_reward = act_seqs[best], reward_seqs[best]
    out = {"act_seq": best_act_seq, "mppi_act_seq": new_act_seq, "best_reward": best_reward, "reward_seqs": reward_seqs,
           "best_model_output": None, "best_eval_output": None}
    if rollout_best:                                                                   # planner.py:268-271
        out["best_model_output"] = model_rollout_fn(state_cur, best_act_seq.unsqueeze(0))
        out["best_eval_output"] = evaluate_traj_fn(out["best_model_output"]["state_seqs"], best_act_seq.unsqueeze(0),
                                                   state_cur=state_cur)
    return out
