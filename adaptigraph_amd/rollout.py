"""One step of the open-loop EVAL rollout on the engine (reference src/dynamics/rollout/rollout.py:108-260, the body of
`rollout_from_start_graph`'s loop): predict -> bounding planes from the prediction -> rule-based graph rebuild with the max_nR
back-off -> history shift (with `store_rest_state` the rest frame stays in slot 0) -> the next step's graph dictionary.

This closes SURVEY §8(f) rank 3: the single-graph builder, its tool rules and the back-off loop (graph.py) were there; this is
the step loop they live in.  What stays out, as in §8: the dataset side of rollout.py (ground-truth lookup, error bookkeeping,
visualisation) - the caller passes the tool keypoints of the next frame pair.

Everything numeric runs on the device: the model forward (ag_forward), the plane bounds (torch reductions; SIX scalars come back to
the host because the reference forms its thresholds from them as Python / numpy scalars, rollout.py:132-139, graph.py:134), the
edge builder and its tool rules (ag_build_edges_single, ag_edges_apply_tool_rule), the history shift.  The back-off's retry
decision (does the graph fit max_nR?) is one integer read per attempt, where the reference catches pad_torch's exception.
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import construct_edges_with_backoff

_GRAPH_KEYS = ("attrs", "p_rigid", "p_instance", "obj_mask", "eef_mask", "state_mask", "material_index")   # rollout.py:242-248


def surface_bounds(obj_kp_vis, ratio):
    """rollout.py:132-139: the six plane bounds of the predicted object keypoints (N_fps, 3), written as the reference writes them
    on numpy float32 scalars (`np.max(...) * ratio` etc., so the products round as they do there under the installed numpy).
    obj_kp_vis may be a device tensor: max / min run where it lives and six floats are read back."""
    if isinstance(obj_kp_vis, torch.Tensor):
        ext = torch.stack([obj_kp_vis.amax(0), obj_kp_vis.amin(0)]).to(torch.float32).cpu().numpy()      # one read-back
    else:
        kp = np.asarray(obj_kp_vis, np.float32)
        ext = np.stack([kp.max(0), kp.min(0)])
    mx, mn = ext[0], ext[1]
    max_y = mx[1] * ratio
    min_y = mn[1]
    max_x = mx[0] * ratio
    max_z = mx[2] * ratio
    min_x = mn[0]
    min_x = (max_x - min_x) * (1 - ratio) + min_x
    min_z = mn[2]
    min_z = (max_z - min_z) * (1 - ratio) + min_z
    return dict(max_y=max_y, min_y=min_y, max_x=max_x, max_z=max_z, min_x=min_x, min_z=min_z)


@torch.no_grad()
def rollout_eval_step(model, graph, eef_kp_start, eef_kp_end, *, adj_thresh, topk, max_nR, connect_tool_all=False,
                      connect_tool_all_non_fixed=True, connect_tool_surface=False, connect_tool_surface_ratio=1.0,
                      knn_thresh=1.0, min_kNN=1.0, knn_increment=0.1, store_rest_state=False, dense=True, trail=None):
    """graph: the batched (B = 1) dictionary `model(**graph)` takes - 'state' (1,n_his,N+M,3), 'action' (1,N+M,3), 'attrs',
    'p_instance', 'obj_mask' (1,N) bool, 'eef_mask', 'state_mask' (1,N+M) bool, '<material>_physics_param', and the edges either
    as dense one-hot 'Rr' / 'Rs' (1,n_rel,N+M) like the reference or as 'edges' (an EdgeList).  eef_kp_start / eef_kp_end (M,3):
    the tool keypoints at the next frame pair (rollout.py:158-161).  The keyword arguments are the dataset config's entries
    rollout.py:27-51 reads.  Returns (new_graph, pred_state (1,N,3), pred_motion): new_graph holds 'Rr' / 'Rs' padded to max_nR
    (dense=True: what rollout.py:237-248 builds, so truncate_graph + model(**graph) run on it unchanged) or 'edges' (dense=False).
    trail (list): receives the back-off's (kNN, topk, n_rel) per attempt, first attempt included."""
    dev = graph["state"].device
    if "edges" in graph:
        fwd = {k: v for k, v in graph.items() if k not in ("Rr", "Rs")}
    else:
        fwd = graph
    pred_state, pred_motion = model(**fwd)                                                # rollout.py:112
    obj_mask = graph["obj_mask"][0].to(dev).to(torch.bool)
    obj_kp = pred_state[0][obj_mask]                                                      # :121, :127 (obj_kp_num = obj_mask.sum())
    bounds = surface_bounds(obj_kp, connect_tool_surface_ratio)                           # :132-139
    eef_start = torch.as_tensor(eef_kp_start, dtype=torch.float32).to(dev)
    eef_end = torch.as_tensor(eef_kp_end, dtype=torch.float32).to(dev)
    n_obj = pred_state.shape[1]
    states = torch.cat([pred_state[0], eef_start], 0).contiguous()                        # :163  (N+M, 3)
    states_delta = torch.zeros_like(states)
    states_delta[n_obj:n_obj + eef_start.shape[0]] = eef_end - eef_start                  # :165-166
    edges = construct_edges_with_backoff(states, adj_thresh, graph["state_mask"][0], graph["eef_mask"][0], topk, max_nR,
                                         knn_thresh=knn_thresh, min_kNN=min_kNN, knn_increment=knn_increment, as_index=not dense,
                                         trail=trail, connect_tools_all=connect_tool_all, connect_tools_surface=connect_tool_surface,
                                         connect_tool_all_non_fixed=connect_tool_all_non_fixed, **bounds)            # :168-222
    hist = graph["state"][0]
    if store_rest_state:                                                                  # :224-229 the rest frame stays
        hist = torch.cat([hist[:1], hist[2:], states[None]], 0)
    else:                                                                                 # :231-232
        hist = torch.cat([hist[1:], states[None]], 0)
    new_graph = {"state": hist[None].to(torch.float32), "action": states_delta[None].to(torch.float32)}
    if dense:
        new_graph["Rr"], new_graph["Rs"] = edges[0][None].to(torch.float32), edges[1][None].to(torch.float32)
    else:
        new_graph["edges"] = edges
    for k in _GRAPH_KEYS:
        if k in graph:
            new_graph[k] = graph[k]
    for k in graph:
        if k.endswith("_physics_param"):                                                  # :249-253
            new_graph[k] = graph[k]
    return new_graph, pred_state, pred_motion


@torch.no_grad()
def rollout_eval(model, graph, eef_pos, start_frame, n_steps, **cfg):
    """n_steps of rollout_eval_step along a tool trajectory eef_pos (T, M, 3): step i uses the frame pair (start_frame + i - 1,
    start_frame + i) the way rollout.py:158-161 looks it up from the next pair (n_future = 1).  -> (final graph, [pred_state (N,3)
    per step], [back-off trail per step]).  The predictions stay on the device; nothing but the step's own decisions is read back."""
    preds, trails = [], []
    for i in range(1, n_steps + 1):
        tr = []
        graph, pred, _ = rollout_eval_step(model, graph, eef_pos[start_frame + i - 1], eef_pos[start_frame + i], trail=tr, **cfg)
        preds.append(pred[0])
        trails.append(tr)
    return graph, preds, trails
