"""dynamics() / dynamics_masked(): the callables the planner binds with functools.partial
(reference src/planning/plan.py:190, src/planning/real_world/planner.py:246,270,
src/planning/physics_param_optimizer.py:219).  Same signatures and return dicts as
src/planning/forward_dynamics.py:12-205 and :209-399.

Two ways in, same engine underneath:
  * actions on the HOST (or option device_decode = 0): this file decodes the action batch and lays out the tool keypoints
    with torch CPU ops, spelled as the reference spells them so every cos/sin/multiply rounds identically (bit-equal
    'action_seqs' against a CPU reference); everything after that - graph build, GNN forward, tool advance, history shift,
    capture - runs inside ONE C-ABI call (ag_rollout) with no host sync per step.
  * actions RESIDENT ON THE GPU (what the planner's sampler produces) and a task config that bounds the push length
    (task_config['action_upper_lim'], planning/*.yaml:28-29): ag_rollout_actions decodes them and plans the launches in a
    device kernel - the host never reads an action, so nothing waits for the GPU between the sampler and the first rollout
    kernel.  cos/sin are then the device's, as they would be for a reference running on a GPU: 'action_seqs' agrees with a
    CPU decode to ~1e-7, not bit for bit.

Deviations, both on paths that do not change any returned value:
  * candidates are advanced only while some candidate of their launch chunk is still live; the reference steps
    the whole batch to the batch maximum of action_repeat and discards the surplus (:156-161).
  * "Exceeds max dims" is raised when a graph that is actually CONSUMED by a forward has more than max_nR edges.
    The reference also pads (and can raise on) the graph it rebuilds after the last repeat, which nothing reads
    (:171-174), and checks the batch max including already-captured candidates.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .context import ptr, current_stream, _require_gpu
from .model import DynamicsPredictor
from .plan_utils import decode_action


def _tool_layout(decoded, theta, task_config):
    """forward_dynamics.py:42-78 / :237-273 for every look-ahead step at once (CPU tensors).
    decoded (B,H,4), theta (B,H) -> eef_xz (B,H,M,2), eef_delta (B,H,M,3)."""
    pts = task_config["pusher_points"]
    ratio = task_config["sim_real_ratio"]
    B, H = theta.shape
    M = len(pts)
    if M not in (1, 5):
        raise NotImplementedError("pusher not implemented")
    xz = torch.zeros((B, H, M, 2))
    delta = torch.zeros((B, H, M, 3))
    delta[..., 0] = (decoded[..., 2] - decoded[..., 0]).unsqueeze(-1)
    delta[..., 2] = (decoded[..., 3] - decoded[..., 1]).unsqueeze(-1)
    xz[:, :, 0, 0] = decoded[..., 0]
    xz[:, :, 0, 1] = decoded[..., 1]
    for k in range(1, M):
        xz[:, :, k, 0] = decoded[..., 0] + float(pts[k][1]) * ratio * torch.sin(theta)
        xz[:, :, k, 1] = decoded[..., 1] - float(pts[k][1]) * ratio * torch.cos(theta)
    return xz, delta


def _physics(ppm_optimizer, physics_param, N_o, dev):
    material_dims = ppm_optimizer.material_dims
    physics_param = ppm_optimizer.physics_param if physics_param is None else physics_param
    assert len(material_dims) == 1          # the model asserts exactly one *_physics_param key (model.py:186-187)
    (name, _dim), = material_dims.items()
    if name not in physics_param:
        return 0.0, None                    # forward_dynamics.py:152-153 zeros
    v = physics_param[name].detach().to("cpu", torch.float32).reshape(-1)
    if v.numel() == 1:
        return float(v[0]), None
    assert v.numel() == N_o                 # model.py:204 reshape(B, n_p, 1)
    return 0.0, v.to(dev).contiguous()


def _run(model, dev, task, ppm_optimizer, physics_param, B, H, N_o, y_mode, state0, obj_mask, xz, delta, repeat,
         sync=True, overflow_flag=None):
    if not isinstance(model, DynamicsPredictor):
        raise TypeError("model must be an adaptigraph_amd.DynamicsPredictor")
    eng = model.engine(dev)
    M = ppm_optimizer.eef_num
    assert xz.shape[2] == M
    assert int(task["n_his"]) == model.n_his, "task_config['n_his'] (forward_dynamics.py:16) must be the model's n_his"
    grip = bool(task["gripper_enable"])
    phys_val, phys_vec = _physics(ppm_optimizer, physics_param, N_o, dev)
    adj = ppm_optimizer.adj_thresh
    p = _lib.AgRolloutParams(B, H, N_o, M, int(task["topk"]), int(bool(task["connect_tools_all"])),
                             int(task["max_nR"]), y_mode, float(adj), float(0.01 * task["sim_real_ratio"]) if grip else 0.0,
                             int(grip), phys_val)
    xz_d = xz.to(torch.float32).contiguous().to(dev)
    delta_d = delta.to(torch.float32).contiguous().to(dev)
    rep = repeat.to("cpu", torch.int32).contiguous()
    out = torch.empty((B, H, N_o, 3), device=dev, dtype=torch.float32)
    args = [eng.ctx, current_stream(dev), C.byref(p), ptr(state0), ptr(obj_mask), ptr(xz_d), ptr(delta_d),
            C.c_void_p(rep.data_ptr()), ptr(phys_vec), ptr(out)]
    if sync:
        eng.check(eng.lib.ag_rollout(*args))
    else:
        eng.check(eng.lib.ag_rollout_async(*args, ptr(overflow_flag)))
    return out


def _repeat_bound(task):
    """Upper bound of action_repeat = int(length) from the task config's action limits (planning/*.yaml:28-29), or None."""
    lim = task.get("action_upper_lim") if hasattr(task, "get") else None
    if lim is None:
        return None
    try:
        return int(float(lim[3]))                       # yaml lists; a GPU tensor here would cost the sync this path avoids
    except (TypeError, IndexError, ValueError):
        return None


def _run_device_actions(model, eng, dev, task, ppm_optimizer, physics_param, action, state0, bound, sync, flags, strict):
    """ag_rollout_actions: decode + launch plan on the device (see module docstring).  Returns (state_seqs, decoded), or None
    when a repeat count beyond `bound` was seen and `strict` is off (the caller then takes the host-decode path, which - like
    the reference, forward_dynamics.py:156 - accepts any action length)."""
    assert int(task["n_his"]) == model.n_his, "task_config['n_his'] (forward_dynamics.py:16) must be the model's n_his"
    B, H = action.shape[0], action.shape[1]
    N_o, M = state0.shape[0], ppm_optimizer.eef_num
    pts = task["pusher_points"]
    if len(pts) != M or M not in (1, 5):
        raise NotImplementedError("pusher not implemented")
    grip = bool(task["gripper_enable"])
    phys_val, phys_vec = _physics(ppm_optimizer, physics_param, N_o, dev)
    p = _lib.AgRolloutParams(B, H, N_o, M, int(task["topk"]), int(bool(task["connect_tools_all"])), int(task["max_nR"]), 0,
                             float(ppm_optimizer.adj_thresh), float(0.01 * task["sim_real_ratio"]) if grip else 0.0,
                             int(grip), phys_val)
    offs = (C.c_float * 8)(*([0.0] + [float(pts[k][1]) * task["sim_real_ratio"] for k in range(1, M)] + [0.0] * (8 - M)))
    act = action.detach().to(dev, torch.float32).contiguous()
    out = torch.empty((B, H, N_o, 3), device=dev, dtype=torch.float32)
    decoded = torch.empty((B, H, 4), device=dev, dtype=torch.float32)
    own_flags = flags is None
    if own_flags:
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
    assert flags.numel() >= 2 and flags.dtype == torch.int32
    eng.check(eng.lib.ag_rollout_actions(eng.ctx, current_stream(dev), C.byref(p), ptr(state0), ptr(act),
                                         float(task["push_length"]), offs, int(bound), ptr(phys_vec), ptr(out), ptr(decoded),
                                         ptr(flags)))
    if sync:
        seen = flags[:2].tolist()                                                       # the one wait of the call
        if seen[0] > int(task["max_nR"]):
            raise Exception("Exceeds max dims")                                        # utils.py:63-65
        if seen[1] > bound:
            if not strict:
                return None
            raise ValueError(f"an action's repeat count {seen[1]} exceeds the task config's action_upper_lim[3] = {bound}: "
                             "the device-planned rollout launches every look-ahead step that many times (pass actions "
                             "within the limits, or leave option device_decode at -1 / 0)")
    else:
        # no wait, so no second chance: a candidate whose repeat exceeds the bound was stepped only `bound` times and never
        # captured.  Its rows must not look like a valid (all-zero) state to a cost function that does not read flags[1]:
        # they become NaN (two small device ops, no sync; flags[1] still reports the count)
        over = (act[..., 3].to(torch.int32) > int(bound)).any(dim=1)
        out.masked_fill_(over[:, None, None, None], float("nan"))
    return out, decoded


@torch.no_grad()
def rollout_work(state, action, model, device, ppm_optimizer, physics_param=None):
    """Model forwards each candidate of `dynamics(state, action, ...)` would be stepped on this engine, WITHOUT rolling anything
    out: (B,) int64 numpy.  With the contact-free prefix in play a candidate counts only the forwards from its first contact on
    (0 if it never touches); otherwise its action_repeat summed over the look-ahead steps.  For cutting work-balanced shards of a
    candidate batch across GPUs (sharding.sharded_candidate_rewards(work_fn=...)): every rank calls it on the FULL batch and gets
    the same numbers - no exchange.  The tool-free base rollout it computes stays in the context for the dynamics() call that
    follows.  Needs GPU-resident (or movable) actions and a task config that bounds the push length; else returns the repeats."""
    import numpy as np
    task = ppm_optimizer.task_config
    dev = _require_gpu(device)
    B, H = action.shape[0], action.shape[1]
    bound = _repeat_bound(task)
    eng = model.engine(dev) if isinstance(model, DynamicsPredictor) else None
    if eng is None or bound is None or not (0 <= bound <= 1024) or ppm_optimizer.eef_num > 8 or eng.get_option("device_decode") == 0:
        _, repeat = decode_action(action.detach().to("cpu", torch.float32), push_length=task["push_length"])
        return repeat.clamp(min=0).sum(1).to(torch.int64).numpy()
    assert int(task["n_his"]) == model.n_his
    state0 = state.detach().to(dev, torch.float32).contiguous()
    N_o, M = state0.shape[0], ppm_optimizer.eef_num
    pts = task["pusher_points"]
    if len(pts) != M or M not in (1, 5):
        raise NotImplementedError("pusher not implemented")
    grip = bool(task["gripper_enable"])
    phys_val, phys_vec = _physics(ppm_optimizer, physics_param, N_o, dev)
    p = _lib.AgRolloutParams(B, H, N_o, M, int(task["topk"]), int(bool(task["connect_tools_all"])), int(task["max_nR"]), 0,
                             float(ppm_optimizer.adj_thresh), float(0.01 * task["sim_real_ratio"]) if grip else 0.0,
                             int(grip), phys_val)
    offs = (C.c_float * 8)(*([0.0] + [float(pts[k][1]) * task["sim_real_ratio"] for k in range(1, M)] + [0.0] * (8 - M)))
    act = action.detach().to(dev, torch.float32).contiguous()
    work = np.zeros(B, np.int32)
    eng.check(eng.lib.ag_rollout_work(eng.ctx, current_stream(dev), C.byref(p), ptr(state0), ptr(act), float(task["push_length"]),
                                      offs, int(bound), ptr(phys_vec), work.ctypes.data_as(C.c_void_p)))
    return work.astype(np.int64)


@torch.no_grad()
def dynamics(state, action, model, device, ppm_optimizer, physics_param=None, _sync=True, _overflow_flag=None):
    """state (N_o,3), action (B,H,4) -> {'state_seqs': (B,H,N_o,3), 'action_seqs': (B,H,4)}"""
    task = ppm_optimizer.task_config
    dev = _require_gpu(device)
    B, H = action.shape[0], action.shape[1]
    if action.is_cuda and isinstance(model, DynamicsPredictor):
        eng = model.engine(dev)
        mode = eng.get_option("device_decode")
        bound = _repeat_bound(task)
        if mode == 1 and bound is None:
            raise ValueError("option device_decode = 1 needs task_config['action_upper_lim'] (the bound of action_repeat)")
        # automatic mode (-1) never narrows the reference's contract: a bound the device plan cannot serve (outside
        # [0, 1024], more than 8 tool points) or - when this call waits for its result anyway - a repeat beyond the bound
        # sends the call down the host-decode path, which steps to any repeat count (forward_dynamics.py:156)
        servable = bound is not None and 0 <= bound <= 1024 and ppm_optimizer.eef_num <= 8
        if mode == 1 or (mode != 0 and servable):
            state0 = state.detach().to(dev, torch.float32).contiguous()
            res = _run_device_actions(model, eng, dev, task, ppm_optimizer, physics_param, action, state0, bound,
                                      _sync, _overflow_flag, strict=(mode == 1))
            if res is not None:
                return {"state_seqs": res[0], "action_seqs": res[1].to(action.device)}
    action_cpu = action.detach().to("cpu", torch.float32)
    decoded, repeat = decode_action(action_cpu, push_length=task["push_length"])          # :23
    xz, delta = _tool_layout(decoded, action_cpu[:, :, 2], task)
    state0 = state.detach().to(dev, torch.float32).contiguous()
    N_o = state0.shape[0]
    out = _run(model, dev, task, ppm_optimizer, physics_param, B, H, N_o, 0, state0, None, xz, delta, repeat,
               sync=_sync, overflow_flag=_overflow_flag)
    return {"state_seqs": out, "action_seqs": decoded.to(action.device)}


@torch.no_grad()
def dynamics_masked(state_init, state_mask, action, model, device, ppm_optimizer, physics_param=None, _sync=True,
                    _overflow_flag=None):
    """state_init (B,max_nobj,3), state_mask (B,max_nobj) bool, action (B,4)
    -> {'state_seqs': (B,max_nobj,3), 'action_seqs': (B,4)}
    _sync=False (r05, the contract dynamics() has): enqueue only; `_overflow_flag` (int32 device tensor, zeroed by the caller)
    receives the largest edge count seen if a graph exceeded max_nR - the caller raises Exception("Exceeds max dims") when it reads
    its results (physics_param_optimizer.dynamics_error_sweep evaluates a whole population / sweep that way).  Pass CPU-resident
    actions and physics parameters to keep the call free of read-backs."""
    task = ppm_optimizer.task_config
    dev = _require_gpu(device)
    B = state_init.shape[0]
    action_cpu = action.detach().to("cpu", torch.float32)[:, None]                         # :218
    decoded, repeat = decode_action(action_cpu, push_length=task["push_length"])
    xz, delta = _tool_layout(decoded, action_cpu[:, :, 2], task)
    state0 = state_init.detach().to(dev, torch.float32).contiguous()
    N_o = state0.shape[1]
    mask_u8 = state_mask.detach().to(dev).to(torch.bool).contiguous().view(torch.uint8)
    out = _run(model, dev, task, ppm_optimizer, physics_param, B, 1, N_o, 1, state0, mask_u8, xz, delta, repeat,
               sync=_sync, overflow_flag=_overflow_flag)
    return {"state_seqs": out[:, 0], "action_seqs": decoded[:, 0].to(action.device)}


@torch.no_grad()
def dynamics_mixed(batches, device, one_stream_each=None, largest_first=True, streams_each=None):
    """One evaluation of a MIXED batch of variable-size graphs (BASELINE configs[4]: rope + granular + cloth candidates, every one
    with its own particle count, graph rebuilt every step): `batches` is a list of per-material argument tuples
        (state_init (B_m,max_nobj_m,3), state_mask (B_m,max_nobj_m), action (B_m,4), model_m, ppm_optimizer_m[, physics_param_m])
    i.e. what one would hand to dynamics_masked (forward_dynamics.py:209-399) material by material, each material with its own model
    (= its own engine context and weights).  -> [dynamics_masked's result dictionary per batch], bit-equal to the sequential calls.

    The materials do not depend on each other, so batch m is enqueued on side stream m without waiting for anything
    (dynamics_masked(_sync=False) on that context's per-stream call slot) and the small graphs' latency-bound launch chains run
    under the large graphs' kernels; the caller's stream waits (on the GPU) for all of them, and ONE read-back brings every
    batch's flags: Exception("Exceeds max dims") (utils.py:63-65) if any graph of any batch outgrew its max_nR.  Pass CPU-resident
    actions / physics parameters to keep the call free of other read-backs.
    one_stream_each: keep every engine on its one side stream instead of letting it fork large batches onto its in-library streams
    (streams_each: that many in-library streams per engine instead of its by-size choice).  largest_first: enqueue the batches in
    descending order of their padded row count, so that the small graphs' launch chains fill the large ones' gaps instead of forming
    the tail (results come back in the order of `batches` either way)."""
    from .context import side_streams
    dev = _require_gpu(device)
    n = len(batches)
    if n == 0:
        return []
    flags = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream(dev)
    entry = torch.cuda.Event()
    entry.record(cur)
    side = side_streams(dev, n)
    pin = bool(one_stream_each)                             # (measured on configs[4]: 167.9 ms with the engines forking by size, 170.9 pinned)
    out = [None] * n
    order = sorted(range(n), key=lambda m: -int(batches[m][0].shape[0]) * int(batches[m][0].shape[1])) if largest_first else list(range(n))
    for m in order:
        batch, st = batches[m], side[m]
        state_init, state_mask, action, model, ppm = batch[:5]
        phys = batch[5] if len(batch) > 5 else None
        st.wait_event(entry)
        with torch.cuda.stream(st):
            if (pin or streams_each) and isinstance(model, DynamicsPredictor):
                with model.engine(dev).options(streams=1 if pin else int(streams_each)):
                    res = dynamics_masked(state_init, state_mask, action, model, dev, ppm, physics_param=phys, _sync=False, _overflow_flag=flags[m])
            else:
                res = dynamics_masked(state_init, state_mask, action, model, dev, ppm, physics_param=phys, _sync=False, _overflow_flag=flags[m])
        done = torch.cuda.Event()
        done.record(st)
        cur.wait_event(done)
        res["state_seqs"].record_stream(cur)
        out[m] = res
    seen = flags.tolist()                                                              # the one wait of the call
    for (batch, (seen_nR, _)) in zip(batches, seen):
        if seen_nR > int(batch[4].task_config["max_nR"]):
            raise Exception("Exceeds max dims")                                        # utils.py:63-65
    return out
