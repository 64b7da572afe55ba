"""CPU ORACLE for the per-candidate cost functions.  TEST INFRASTRUCTURE ONLY (see adaptigraph_oracle.py header).

numpy restatement of reference src/planning/losses.py:4-92 and running_cost (src/planning/plan.py:27-59).
Pinned against tests/golden/costs.npz, generated from the real reference by tests/golden/make_golden.py --costs
(tests/test_oracle_vs_golden.py::test_costs_*).  Float path: tolerance, not bit-exactness (exp / sqrt / sum order).
"""
import numpy as np

F32 = np.float32


def _norm_last(d):
    return np.sqrt((d * d).sum(-1, dtype=F32)).astype(F32)


def chamfer(x, y):
    """losses.py:4-10.  x (B,N,D), y (By,M,D) with By in {1,B} -> (B,)"""
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    dis = _norm_last(x[:, None, :, :] - y[:, :, None, :])            # (B, M, N)
    dis_xy = dis.min(2).mean(1, dtype=F32)                           # :7 for every y point the nearest x, mean over M
    dis_yx = dis.min(1).mean(1, dtype=F32)                           # :8 for every x point the nearest y, mean over N
    return (dis_xy + dis_yx).astype(F32)


def mean_chamfer(state_pred, state_real, pred_mask, real_mask):
    """losses.py:12-24 (returns float64 like np.array of python floats)."""
    out = []
    for i in range(state_pred.shape[0]):
        out.append(float(chamfer(state_pred[i][pred_mask[i]][None], state_real[i][real_mask[i]][None])[0]))
    return np.array(out)


def box_loss(state, target):
    """losses.py:26-35.  state (B,N,3), target (2,2) [[xmin,xmax],[zmin,zmax]] -> (B,)"""
    state = np.asarray(state, F32)
    t = np.asarray(target, F32)
    x, z = state[:, :, 0], state[:, :, 2]
    xd = np.maximum(t[0, 0] - x, 0) + np.maximum(x - t[0, 1], 0)
    zd = np.maximum(t[1, 0] - z, 0) + np.maximum(z - t[1, 1], 0)
    return np.sqrt(xd * xd + zd * zd).mean(1, dtype=F32).astype(F32)


def _state_2d(state_pred, state_init):
    """losses.py:42-43 / :83-84: look-ahead step h is judged against the cloud BEFORE it: init, then pred[:, :-1]."""
    B = state_pred.shape[0]
    init = np.broadcast_to(state_init[None, None][..., [0, 2]], (B, 1) + state_init[:, [0, 2]].shape)
    return np.concatenate([init, state_pred[:, :-1][..., [0, 2]]], 1).astype(F32)


def rope_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:37-48 -> (B,H)"""
    state_pred, action, state_init = (np.asarray(a, F32) for a in (state_pred, action, state_init))
    pt = action[:, :, :2]
    d = _norm_last(pt[:, :, None, :] - _state_2d(state_pred, state_init)).min(-1)
    d = np.maximum(d - F32(0.02 * sim_real_ratio), 0)
    return np.exp(-d * F32(100.0)).astype(F32)


def cloth_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:50-64 -> (B,H).  NB: uses only state_init, and normalises by the BATCH-GLOBAL max (:62)."""
    action, state_init = np.asarray(action, F32), np.asarray(state_init, F32)
    pt = action[:, :, :2]
    d = _norm_last(pt[:, :, None, :] - state_init[None, None][..., [0, 2]])
    dmin = np.maximum(d.min(-1) - F32(0.005 * sim_real_ratio), 0)
    dmax = np.minimum(d.max(-1), F32(0.4 * sim_real_ratio))
    dmax = dmax / dmax.max()
    return (F32(1.0) - np.exp(-dmin * F32(100.0)) - dmax * F32(0.2)).astype(F32)


def cloth_penalty_terms(state_pred, action, state_init, sim_real_ratio=10.0):
    """The two per-(candidate, step) terms of losses.py:50-64 BEFORE the batch-global normalisation (:62) - what a rank
    holding a shard of the batch can compute on its own: [exp(-dmin*100), dmax].  cloth_penalty = 1 - t0 - 0.2*t1/max(t1)."""
    action, state_init = np.asarray(action, F32), np.asarray(state_init, F32)
    pt = action[:, :, :2]
    d = _norm_last(pt[:, :, None, :] - state_init[None, None][..., [0, 2]])
    dmin = np.maximum(d.min(-1) - F32(0.005 * sim_real_ratio), 0)
    dmax = np.minimum(d.max(-1), F32(0.4 * sim_real_ratio))
    return np.stack([np.exp(-dmin * F32(100.0)), dmax], -1).astype(F32)


def granular_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:66-92 -> (B,H): 9 points along the pusher blade."""
    state_pred, action, state_init = (np.asarray(a, F32) for a in (state_pred, action, state_init))
    x, z, th = action[:, :, 0], action[:, :, 1], action[:, :, 2]
    r = F32(0.05 * sim_real_ratio)
    import torch
    tth = torch.from_numpy(np.ascontiguousarray(th))
    dx = (r * torch.sin(tth)).numpy()
    dz = (-r * torch.cos(tth)).numpy()
    pts = np.stack([np.stack([x + F32(c) * dx, z + F32(c) * dz], -1) if c >= 0 else
                    np.stack([x - F32(-c) * dx, z - F32(-c) * dz], -1)
                    for c in (-1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75, 1.0)], 2)     # (B,H,9,2)
    s2 = _state_2d(state_pred, state_init)                                                  # (B,H,N,2)
    d = _norm_last(pts[:, :, :, None, :] - s2[:, :, None, :, :]).min(-1).min(-1)
    d = np.maximum(d - F32(0.02 * sim_real_ratio), 0)
    return np.exp(-d * F32(100.0)).astype(F32)


def running_cost(state, action, state_cur, error_func, penalty_func, bbox):
    """plan.py:27-59 -> reward (B,)"""
    state = np.asarray(state, F32)
    B, H = state.shape[:2]
    error = error_func(state.reshape(B * H, state.shape[2], 3)).reshape(B, H)           # :35-36
    error_weight = F32(2.0) / (error.max() + F32(1e-6))                                # :37 batch-global max
    pen = penalty_func(state, action, state_cur)                                        # :39
    xmax, xmin = state[..., 0].max(2), state[..., 0].min(2)                             # :41-44
    zmax, zmin = state[..., 2].max(2), state[..., 2].min(2)
    bb = np.asarray(bbox, np.float64)
    box = np.stack([np.maximum(xmin - F32(bb[0, 0]), 0), np.maximum(F32(bb[0, 1]) - xmax, 0),
                    np.maximum(zmin - F32(bb[1, 0]), 0), np.maximum(F32(bb[1, 1]) - zmax, 0)], -1)   # :45-50
    box = np.exp(-box * F32(100.0)).max(-1)                                             # :51
    return (-error_weight * error[:, -1] - F32(5.0) * pen.mean(1, dtype=F32) - F32(5.0) * box.mean(1, dtype=F32)).astype(F32)


def dynamics_error(W, pstep, physics_value, task, state_init_list, state_real_list, actions):
    """physics_param_optimizer.py:178-226 with the oracle's dynamics_masked and mean_chamfer."""
    from oracle import adaptigraph_oracle as O
    n, max_nobj = len(actions), task["max_nobj"]
    init = np.zeros((n, max_nobj, 3), F32); fin = np.zeros((n, max_nobj, 3), F32)
    im = np.zeros((n, max_nobj), bool); fm = np.zeros((n, max_nobj), bool)
    for i in range(n):
        ni, nf = state_init_list[i].shape[0], state_real_list[i].shape[0]
        init[i, :ni], fin[i, :nf] = state_init_list[i], state_real_list[i]
        im[i, :ni], fm[i, :nf] = True, True
    out = O.dynamics_masked(W, pstep, init, im, np.stack(actions, 0), task, physics_param=physics_value)
    return float(mean_chamfer(out["state_seqs"], fin, im, fm).mean())
