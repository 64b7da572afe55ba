"""Shared helpers for the parity tests (fixture loading, golden-step indexing)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def task_of(npz):
    return json.loads(bytes(npz["task_json"]).decode())


def split_edges(npz, prefix):
    cnt = npz[prefix + "n_edges"]
    off = np.concatenate([[0], np.cumsum(cnt)])
    recv, send = npz[prefix + "recv"], npz[prefix + "send"]
    return [(recv[off[b]:off[b + 1]], send[off[b]:off[b + 1]]) for b in range(len(cnt))]


def golden_step_index(repeat):
    """The reference steps the whole batch to max(repeat[:, li]) per look-ahead step.
    Returns base[li] = index of the first recorded forward of look-ahead step li."""
    repeat = np.atleast_2d(repeat)
    mx = repeat.max(0)
    return np.concatenate([[0], np.cumsum(mx)])[:-1]


# ---- closed-form stand-ins for the two callables a Planner is configured with (planner.py:47-61); shared by
# tests/golden/make_golden.py --planner (which drives the REFERENCE Planner with them) and tests/test_planner.py
def toy_rollout(state_cur, act_seqs):
    """state_cur (P,3), act_seqs (S,H,A) -> {'state_seqs': (S,H,P,3)}: the cloud shifted by the running sum of the first
    two action components (x, z) and tilted by the third, per candidate - independent of the rest of the batch."""
    import torch
    shift = torch.cumsum(act_seqs[..., :2], dim=1)                                   # (S,H,2)
    s = state_cur[None, None].repeat(act_seqs.shape[0], act_seqs.shape[1], 1, 1).clone()
    s[..., 0] += shift[..., 0, None]
    s[..., 2] += shift[..., 1, None]
    if act_seqs.shape[-1] > 2:
        s[..., 1] += 0.1 * torch.sin(act_seqs[..., 2])[..., None] * state_cur[None, None, :, 0]
    return {"state_seqs": s, "action_seqs": act_seqs}


def toy_cost(state_seqs, act_seqs, state_cur=None, weights=None, **kw):
    """-> {'reward_seqs': (S,)}: distance of the final cloud centre to a target, scaled by the batch maximum (the
    batch-global normalisation of running_cost, plan.py:37) plus an action-length term."""
    import torch
    target = torch.tensor([0.4, 0.0, -0.3])
    err = (state_seqs.mean(2) - target).norm(dim=-1)                                   # (S,H)
    w = 2.0 / (err.max() + 1e-6)
    return {"reward_seqs": -w * err[:, -1] - 0.05 * act_seqs.abs().sum((1, 2))}


def toy_planner_config(rollout, cost, action_dim=3, n_sample=16, n_update_iter=3, noise_type="normal"):
    import torch
    return {"action_dim": action_dim, "model_rollout_fn": rollout, "evaluate_traj_fn": cost, "n_sample": n_sample,
            "n_look_ahead": 3, "n_update_iter": n_update_iter, "reward_weight": 20.0,
            "action_lower_lim": torch.tensor([-0.5, -0.4, -1.0][:action_dim]),
            "action_upper_lim": torch.tensor([0.5, 0.6, 1.0][:action_dim]),
            "planner_type": "MPPI", "device": "cpu", "verbose": False, "noise_type": noise_type, "noise_level": 0.2,
            "rollout_best": True}


# ---- full-size fixtures recorded from the reference (tests/golden/make_golden.py --fullsize)
def fullsize_records(g, task):
    """-> per candidate: (records, capture, want).  records[f] = what the reference's forward f saw and produced
    ('recv', 'send' int32, 'state_last' (N,3) = the positions its edge builder was fed, 'pred_pos' (N_o,3)); capture =
    indices of the forwards whose prediction is captured into state_seqs; want = the state_seqs rows of the candidate."""
    import hashlib
    F, B = int(g["n_steps"]), g["action"].shape[0]
    cnt, e16, sha = g["n_edges"], g["edges_i16"], g["edges_sha256"]
    off = np.concatenate([[0], np.cumsum(cnt.ravel())])
    rep = g["action"][..., 3].astype(np.int32)                       # plan_utils.py:16 (truncation)
    assert (rep == rep[0]).all(), "the fixture's candidates share their repeats (forward f is the same step for all)"
    H = rep.shape[1]
    starts = np.concatenate([[0], np.cumsum(rep[0])])                # first forward of every look-ahead step
    out = []
    for b in range(B):
        recs = []
        for f in range(F):
            k = f * B + b
            pr = e16[off[k]:off[k + 1]].astype(np.int32)
            r, s = np.ascontiguousarray(pr[:, 0]), np.ascontiguousarray(pr[:, 1])
            assert hashlib.sha256(r.tobytes() + s.tobytes()).digest() == sha[f, b].tobytes(), (f, b)
            li = int(np.searchsorted(starts, f, side="right") - 1)
            if f == starts[li]:
                start = g["state_init"][b] if "state_init" in g.files else g["state0"]      # masked variant: per-candidate clouds
                obj = start if li == 0 else g["state_seqs"][b, li - 1]
            else:
                obj = g["pred_pos"][f - 1, b]
            recs.append({"recv": r, "send": s, "pred_pos": g["pred_pos"][f, b],
                         "state_last": np.concatenate([obj, g["tool_pos"][f, b]], 0).astype(np.float32)})
        out.append((recs, (starts[1:] - 1).tolist(), [g["state_seqs"][b, li] for li in range(H)]))
    return out
