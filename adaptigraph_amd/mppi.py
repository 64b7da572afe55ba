"""MPPI sampling / update on the HIP engine, behind the reference's function names (src/planning/plan_utils.py:31-101),
and one whole MPC iteration (src/planning/real_world/planner.py:234-277).

The arithmetic lives in csrc/ag_mppi.hip (ag_mppi_sample / ag_mppi_update / ag_mppi_clip).  This module only draws
the random numbers - with the same torch calls, shapes and order as the reference, so a caller that seeds torch gets
the reference's draws - and hands device pointers to the C-ABI.  There is no CPU path.

What the engine changes upstream: the reference evaluates n_sample = 20000 in 40 host-side chunks of 500
(plan.py:177-182, 241-247) because its dense rollout does not fit.  `mpc_iteration` here takes the whole batch in one call
(the engine chunks on the device), optionally sharded over the ranks of a torch.distributed group - with the cost
normalisation over the whole batch.  The reference's chunk-local semantics are kept by planner.Planner
(trajectory_optimization_chunked), which is what a drop-in for plan.py uses.
"""
from __future__ import annotations

import torch

from .context import default_engine, ptr, current_stream, _require_gpu


def _dev_f32(t, dev):
    return torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).contiguous()


def _limits(lo, hi, dev):
    lo, hi = _dev_f32(lo, dev).reshape(-1), _dev_f32(hi, dev).reshape(-1)
    assert lo.numel() == 4 and hi.numel() == 4
    return lo, hi


def clip_actions(action, action_lower_lim, action_upper_lim):
    """plan_utils.py:35-39: theta wrapped into [-pi, pi) (angle_normalize, :31-32), every component clamped."""
    dev = _require_gpu(action.device)
    eng = default_engine(dev)
    assert action.shape[-1] == 4
    a = _dev_f32(action, dev)
    lo, hi = _limits(action_lower_lim, action_upper_lim, dev)
    out = torch.empty_like(a)
    if a.numel():
        eng.check(eng.lib.ag_mppi_clip(eng.ctx, current_stream(dev), ptr(a), ptr(lo), ptr(hi), a.numel() // 4, ptr(out)))
    return out


def angle_normalize(x):
    """plan_utils.py:31-32 on a tensor of angles (any shape)."""
    dev = _require_gpu(x.device)
    a = torch.zeros(x.shape + (4,), device=dev, dtype=torch.float32)
    a[..., 2] = x
    inf = torch.full((4,), float("inf"), device=dev)
    return clip_actions(a, -inf, inf)[..., 2]


_BETA = {}


def _beta(H, dev):
    """plan_utils.py:62 `beta = 0.1 * (10 ** i)` per look-ahead step, resident on the device: a constant, uploaded once.  (Built per
    call it is a pageable host-to-device copy, which blocks the host until everything queued on the stream has run - once per update
    iteration of a planner call, random_interact.py's configuration: 4 x ~10 ms of host time per call in the r06 trace.)"""
    key = (H, dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _BETA:
        _BETA[key] = torch.tensor([0.1 * (10 ** i) for i in range(H)], dtype=torch.float32).to(dev)
    return _BETA[key]


def sample_action_seq(act_seq, action_lower_lim, action_upper_lim, n_sample, device, iter_index=0, noise_level=0.3,
                      push_length=0.10, _draws=None):
    """plan_utils.py:42-77 -> (n_sample, n_look_ahead, 4).  iter_index 0: uniform resampling inside the limits; else
    Gaussian perturbation of the nominal push's start / end points (sample 0 keeps the nominal action).
    `_draws`: the random numbers to use instead of drawing them (tests: the reference's recorded draws)."""
    dev = _require_gpu(device)
    eng = default_engine(dev)
    H = act_seq.shape[0]
    assert act_seq.shape[-1] == 4
    lo, hi = _limits(action_lower_lim, action_upper_lim, dev)
    out = torch.empty((n_sample, H, 4), device=dev, dtype=torch.float32)
    if iter_index == 0:
        u = torch.rand((n_sample, H, 4), device=dev) if _draws is None else _dev_f32(_draws, dev)     # :49
        assert u.shape == (n_sample, H, 4)
        eng.check(eng.lib.ag_mppi_sample(eng.ctx, current_stream(dev), None, ptr(lo), ptr(hi), ptr(u), None, n_sample, H,
                                         0, float(push_length), ptr(out)))
        return out
    if _draws is None:                                                                             # :60, one draw per step
        noise = torch.stack([torch.normal(0, noise_level, (n_sample, 4), device=dev) for _ in range(H)])
    else:
        noise = _dev_f32(_draws, dev)
    assert noise.shape == (H, n_sample, 4)
    scale = _beta(H, dev)                                                                           # :62
    nominal = _dev_f32(act_seq, dev)
    eng.check(eng.lib.ag_mppi_sample(eng.ctx, current_stream(dev), ptr(nominal), ptr(lo), ptr(hi), ptr(noise), ptr(scale),
                                     n_sample, H, 1, float(push_length), ptr(out)))
    return out


def optimize_action_mppi(act_seqs, reward_seqs, reward_weight=100.0, action_lower_lim=None, action_upper_lim=None,
                         push_length=0.10):
    """plan_utils.py:80-101: softmax(reward * reward_weight)-weighted mean of the candidates' start and end points,
    re-encoded as (x, z, theta, length) and limited -> (n_look_ahead, 4)."""
    dev = _require_gpu(act_seqs.device)
    eng = default_engine(dev)
    B, H = act_seqs.shape[0], act_seqs.shape[1]
    assert act_seqs.shape[-1] == 4 and reward_seqs.shape == (B,)
    acts, rew = _dev_f32(act_seqs, dev), _dev_f32(reward_seqs, dev)
    lo, hi = _limits(action_lower_lim, action_upper_lim, dev)
    out = torch.empty((H, 4), device=dev, dtype=torch.float32)
    eng.check(eng.lib.ag_mppi_update(eng.ctx, current_stream(dev), ptr(acts), ptr(rew), ptr(lo), ptr(hi), B, H,
                                     float(reward_weight), float(push_length), ptr(out)))
    return out


@torch.no_grad()
def mpc_iteration(state_cur, act_seq, model_rollout_fn, evaluate_traj_fn, action_lower_lim, action_upper_lim, n_sample,
                  device, noise_level=1.0, reward_weight=500.0, push_length=0.10, iter_index=0, rollout_best=True,
                  act_seqs=None, group=None, n_update_iter=1, reuse_best_rollout=False):
    """Planner.trajectory_optimization_mppi (planner.py:234-277): per update iteration sample -> rollout -> evaluate ->
    MPPI update; the best candidate over all iterations is kept and (optionally) rolled out once more.  plan.py:199 runs
    it with n_update_iter = 1.

    model_rollout_fn(state_cur, act_seqs) and evaluate_traj_fn(state_seqs, act_seqs, state_cur=...) are the same
    partials plan.py builds (plan.py:175, 190).  `group`: shard the candidates over the ranks of that group; every
    rank must call with the same generator state (or pass the same `act_seqs`, which then serves the first iteration).
    The whole candidate batch goes through one rollout call (the engine chunks on the device) and the cost maxima are
    taken over the whole batch; the reference's loop over n_sample / n_sample_chunk chunks with its chunk-local maxima
    and merge_res (plan.py:241-247) is planner.Planner.trajectory_optimization_chunked.
    reuse_best_rollout: take the best candidate's rollout out of the batch it was sampled in instead of rolling it out again
    with a batch of one (exact on this engine - a candidate's rollout does not depend on its batch; single rank only)."""
    from .sharding import shard_bounds, all_gather_costs
    import torch.distributed as dist
    pg = None if group in (None, True) else group
    world = dist.get_world_size(pg) if group is not None else 1
    rank = dist.get_rank(pg) if group is not None else 0
    best_act_seq = best_reward = reward_seqs = best_rows = None
    nominal = act_seq
    for it in range(n_update_iter):
        if act_seqs is None or it > 0:
            act_seqs = sample_action_seq(nominal, action_lower_lim, action_upper_lim, n_sample, device,
                                         iter_index=iter_index + it, noise_level=noise_level, push_length=push_length)
        lo, hi = shard_bounds(act_seqs.shape[0], world, rank)
        model_out = model_rollout_fn(state_cur, act_seqs[lo:hi])
        eval_out = evaluate_traj_fn(model_out["state_seqs"], act_seqs[lo:hi], state_cur=state_cur)
        reward_seqs = eval_out["reward_seqs"]
        if world > 1:
            reward_seqs = all_gather_costs(reward_seqs.contiguous(), act_seqs.shape[0], pg)
        nominal = optimize_action_mppi(act_seqs, reward_seqs, reward_weight, action_lower_lim, action_upper_lim, push_length)
        top = torch.argmax(reward_seqs)
        sel = top.reshape(1)                                                  # (index_select: no read-back of `top`)
        top_reward = torch.index_select(reward_seqs, 0, sel)[0]
        if best_reward is None or bool(top_reward > best_reward):             # planner.py:254-260
            best_act_seq, best_reward = torch.index_select(act_seqs, 0, sel)[0], top_reward
            if reuse_best_rollout and world == 1:
                n = act_seqs.shape[0]
                best_rows = {k: (torch.index_select(v, 0, top.reshape(1)) if isinstance(v, torch.Tensor) and v.dim() > 0 and
                                 v.shape[0] == n else v) for k, v in model_out.items()}
    out = {"act_seq": best_act_seq, "mppi_act_seq": nominal, "best_reward": best_reward, "reward_seqs": reward_seqs,
           "best_model_output": None, "best_eval_output": None}
    if rollout_best:                                                                   # planner.py:268-271
        out["best_model_output"] = best_rows if best_rows is not None else model_rollout_fn(state_cur, best_act_seq.unsqueeze(0))
        out["best_eval_output"] = evaluate_traj_fn(out["best_model_output"]["state_seqs"], best_act_seq.unsqueeze(0),
                                                   state_cur=state_cur)
    return out
