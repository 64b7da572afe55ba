"""CPU ORACLE for the AdaptiGraph GNN-dynamics rollout path.  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy restatement of the reference algorithm.  It is the
checker for the HIP path; it is never the thing shipped or measured.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product package (adaptigraph_amd/) must never import this module.

Parity pin: the reference holds NO golden vectors or tests for this path
(SURVEY.md §4).  The oracle is pinned instead against outputs of the reference
itself, generated in the build container by tests/golden/make_golden.py and
committed as tests/golden/*.npz (tests/test_oracle_vs_golden.py: edge indices
bit-exact, positions within 2e-6 per step).

Each function cites the reference file:line it follows (paths relative to the
reference repo root).

Formulation notes
  * The reference represents edges as dense one-hot matrices Rr/Rs (B,E,N) and
    gathers/scatters with bmm.  A one-hot bmm is an exact gather, so the oracle
    uses int32 (recv, send) index lists - identical values, no N*E memory.
  * The scatter Rr^T.bmm(effect_rel) (model.py:324) is a sum whose order is
    whatever the BLAS picks; the oracle sums in edge order with np.add.at.
    This (and BLAS blocking inside the Linear layers) is why positions are
    compared with a tolerance while edge indices are compared bit-for-bit.
  * decode_action's cos/sin (plan_utils.py:11-20) are evaluated with torch CPU
    ops, like the reference and like the product's host shim, so that the tool
    trajectory bits are identical on both sides of every comparison.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BIG = F32(1e10)

WEIGHT_KEYS = [
    "particle_encoder.model.0", "particle_encoder.model.2", "particle_encoder.model.4",
    "relation_encoder.model.0", "relation_encoder.model.2", "relation_encoder.model.4",
    "particle_propagator.linear", "relation_propagator.linear",
    "non_rigid_predictor.linear_0", "non_rigid_predictor.linear_1", "non_rigid_predictor.linear_2",
]


class TopkTie(Exception):
    """A tie at the k-th boundary inside the radius: the reference's answer is implementation-defined."""


# --------------------------------------------------------------------------- edges
def pairwise_dis(pos):
    """graph.py:251-252  dis = sum((s_r - s_s)**2, -1), fp32, evaluated ((dx^2 + dy^2) + dz^2), no FMA."""
    pos = np.ascontiguousarray(pos, dtype=F32)
    d = pos[:, None, :] - pos[None, :, :]
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def construct_edges_single(pos, thr, mask, tool_mask, topk, connect_tools_all, check_ties=False):
    """One batch element of construct_edges_from_states_batch (graph.py:233-298).

    pos (N,3) f32, thr python float or f32 scalar, mask/tool_mask (N,) bool.
    Returns (recv, send) int32 arrays in the reference's row-major nonzero order
    (sorted by recv, then send) == CSR by receiver.
    """
    N = pos.shape[0]
    thr = F32(thr)
    thr2 = F32(thr * thr)                                   # :248-250 fp32 square of an fp32 threshold
    dis = pairwise_dis(pos)                                 # :251-252
    mask = np.asarray(mask, bool)
    tool = np.asarray(tool_mask, bool)
    dis[~(mask[:, None] & mask[None, :])] = BIG             # :253-256
    dis[tool[:, None] & tool[None, :]] = BIG                # :257-260
    adj = (dis - thr2) < 0                                  # :267
    k = min(N, int(topk))                                   # :270
    if k < N:                                               # :271-274 (k == N keeps everything)
        kth = np.partition(dis, k - 1, axis=1)[:, k - 1][:, None]
        less = dis < kth
        eq = dis == kth
        need = k - less.sum(1, keepdims=True)
        # ties: lowest sender index first == (distance, index) lexicographic order
        sel = less | (eq & (np.cumsum(eq, axis=1) <= need))
        if check_ties:
            tied_in_radius = (eq.sum(1, keepdims=True) > need) & (kth < thr2)
            if tied_in_radius.any():
                raise TopkTie("tie at the k-th boundary inside the radius")
        adj &= sel
    if connect_tools_all:                                   # :276-286
        pad_tool_1 = tool[:, None] & ~tool[None, :]         # :265 tool receiver, non-tool sender
        flag = bool(adj[pad_tool_1].any())                  # :277 batch_mask
        m1 = tool[:, None] & mask[None, :]                  # :262 obj_tool_mask_1
        m2 = tool[None, :] & mask[:, None]                  # :263 obj_tool_mask_2
        adj[m1] = False                                     # :283 / :285
        adj[m2] = flag                                      # :284 / :286
    recv, send = np.nonzero(adj)                            # :293 row-major
    return recv.astype(np.int32), send.astype(np.int32)


def selection_margin(pos, thr, mask, tool_mask, topk):
    """How far (in fp32 squared-distance units) the positions are from changing the graph of construct_edges_single:
    the smallest of (a) |dis - thr^2| over unmasked, non tool-tool pairs (a pair entering / leaving the radius) and
    (b) d_(k+1) - d_(k) over receiver rows whose k-th and (k+1)-th nearest senders are both inside the radius (two
    senders swapping places at the top-k boundary).  A position perturbation e moves dis by about 2*sqrt(dis)*2e, so a
    margin below ~4*thr*tol means the reference's own edge choice is not determined to within a position tolerance tol:
    free-running comparisons past such a step are comparisons of two different, equally valid graphs."""
    N = pos.shape[0]
    thr2 = F32(F32(thr) * F32(thr))
    dis = pairwise_dis(pos)
    mask = np.asarray(mask, bool)
    tool = np.asarray(tool_mask, bool)
    dead = ~(mask[:, None] & mask[None, :]) | (tool[:, None] & tool[None, :])
    radius = np.abs(np.where(dead, np.inf, dis) - thr2).min() if (~dead).any() else np.inf
    swap = np.inf
    k = min(N, int(topk))
    if k < N:
        d = np.where(dead, BIG, dis)
        part = np.partition(d, (k - 1, k), axis=1)
        both_in = part[:, k] < thr2
        if both_in.any():
            swap = (part[both_in, k] - part[both_in, k - 1]).min()
    return float(min(radius, swap))


def _flat_smallest(values, k):
    """torch.topk(values, k, largest=False) membership on a flat vector; ties resolved (value, index) lexicographic
    (torch's own choice is implementation-defined; fixtures hold no such tie)."""
    keep = np.zeros(values.shape[0], bool)
    if k > 0:
        keep[np.argsort(values, kind="stable")[:k]] = True
    return keep


PLANES = ["max_y", "min_x", "max_x", "min_z", "max_z"]      # graph.py:38 order, which np.argsort ties fall back on


def _plane_masks(name, pos, max_y, max_x, max_z, min_x, min_z):
    """graph.py:45-66: per-particle side of the plane test (the scalar is rounded to fp32 by torch's comparison)."""
    axis, bound, ge = {"max_y": (1, max_y, True), "max_x": (0, max_x, True), "max_z": (2, max_z, True),
                       "min_x": (0, min_x, False), "min_z": (2, min_z, False)}[name]
    return pos[:, axis] >= F32(bound) if ge else pos[:, axis] <= F32(bound)


def construct_edges_from_states(pos, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, max_y=None, min_y=None,
                                max_x=None, max_z=None, min_x=None, min_z=None, connect_tools_surface=False,
                                connect_tool_all_non_fixed=True, kNN=1.0, check_ties=False, trace=None):
    """The SINGLE-graph builder (graph.py:68-231).  Differs from the batch builder in two reproduced ways: the
    threshold is squared in Python double precision and only then meets the fp32 distances (:86,101), and
    connect_tools_all is unconditional with tool<->tool removed (:119-122).  With max_y/min_y given the
    'tool to all non-fixed particles' rule (:125-171, optional flat kNN filter) applies, with all five bounds and
    connect_tools_surface the 'tool to the two closest surface planes' rule (:173-221).  `trace` (dict) receives the
    intermediate decisions (check values, chosen planes, keepK)."""
    pos = np.ascontiguousarray(pos, dtype=F32)
    N = pos.shape[0]
    thr2 = F32(float(adj_thresh) * float(adj_thresh))               # :86 double product, rounded by the fp32 subtraction
    dis = pairwise_dis(pos)                                         # :87-88
    mask = np.asarray(mask, bool)
    tool = np.asarray(tool_mask, bool)
    dis[~(mask[:, None] & mask[None, :])] = BIG                     # :89-92
    t12 = tool[:, None] & tool[None, :]
    dis[t12] = BIG                                                  # :93-96
    m1 = tool[:, None] & mask[None, :]                              # :98 tool receiver, valid sender
    m2 = tool[None, :] & mask[:, None]                              # :99 valid receiver, tool sender
    adj = (dis - thr2) < 0                                          # :101
    k = min(N, int(topk))                                           # :109
    if k < N:
        kth = np.partition(dis, k - 1, axis=1)[:, k - 1][:, None]
        less, eq = dis < kth, dis == kth
        need = k - less.sum(1, keepdims=True)
        if check_ties and ((eq.sum(1, keepdims=True) > need) & (kth < thr2)).any():
            raise TopkTie("tie at the k-th boundary inside the radius")
        adj &= less | (eq & (np.cumsum(eq, axis=1) <= need))
    if connect_tools_all:                                           # :119-122
        adj[m1] = False
        adj[m2] = True
        adj[t12] = False
    tr = trace if trace is not None else {}
    if connect_tool_all_non_fixed and max_y is not None and min_y is not None:      # :125
        check = int(adj[m2].sum())                                  # :128-129
        threshold = (max_y - min_y) * 0.1 + min_y                   # :134 caller's own scalar types
        tr["nonfixed_check"] = check
        if check > 0:
            surf = (pos[:, 1] > F32(threshold)) & mask              # :138-143 (fp32 comparison)
            s1 = tool[:, None] & surf[None, :]                      # :144 tool receiver, non-fixed sender
            s2 = tool[None, :] & surf[:, None]                      # :145 non-fixed receiver, tool sender
            count = int(s2.sum())                                   # :151
            tr["nonfixed_n"] = int(surf.sum())
            adj[s1] = False                                         # :153
            adj[s2] = True                                          # :154
            if kNN < 1.0 and kNN > 0.0:                             # :156-169 flat k-nearest filter
                keepK = int(kNN * count)
                tr["keepK"] = keepK
                adj[s2] = adj[s2] & _flat_smallest(dis[s2], keepK)
            adj[t12] = False                                        # :170
    if connect_tools_surface and None not in (max_y, max_x, min_x, max_z, min_z):   # :173
        sel = adj[m2]                                               # :178 flat 0/1 vector
        check = int(sel.sum())
        tr["surface_check"] = check
        if check > 0:
            # :190-194 index s_receiv with the 0/1 VALUES of that vector: every entry selects particle 0 or 1,
            # broadcast over N senders.  Reproduced as written.
            n1 = check
            n0 = sel.shape[0] - check
            def plane_dist(axis, bound):
                d0 = (float(pos[0, axis]) - float(F32(bound))) ** 2
                d1 = (float(pos[1, axis]) - float(F32(bound))) ** 2
                return N * (n0 * d0 + n1 * d1)
            values = [plane_dist(1, max_y), plane_dist(0, min_x), plane_dist(0, max_x), plane_dist(2, min_z),
                      plane_dist(2, max_z)]                         # :36-37 order
            order = np.argsort(values)                              # :39
            first, second = PLANES[order[0]], PLANES[order[1]]
            tr["planes"] = (first, second)
            c1 = _plane_masks(first, pos, max_y, max_x, max_z, min_x, min_z)
            c2 = _plane_masks(second, pos, max_y, max_x, max_z, min_x, min_z)
            surf = c1 & c2 & mask                                   # :201-207
            tr["surface_n"] = int(surf.sum())
            s1 = tool[:, None] & surf[None, :]
            s2 = tool[None, :] & surf[:, None]
            adj[s1] = False                                         # :216
            adj[s2] = True                                          # :217
            adj[t12] = False                                        # :218
    recv, send = np.nonzero(adj)                                    # :225
    return recv.astype(np.int32), send.astype(np.int32)


def construct_edges_batch(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, check_ties=False):
    """graph.py:233-298 for a batch.  adj_thresh: python float or (B,) array.  Returns list of (recv, send)."""
    states = np.asarray(states, F32)
    B = states.shape[0]
    thr = np.full(B, F32(adj_thresh), F32) if np.isscalar(adj_thresh) else np.asarray(adj_thresh, F32)
    return [construct_edges_single(states[b], thr[b], mask[b], tool_mask[b], topk, connect_tools_all, check_ties)
            for b in range(B)]


# --------------------------------------------------------------------------- model
def _linear(x, W, b):
    return x @ W.T + b


def _relu(x):
    return np.maximum(x, F32(0))


def _encoder(x, W, pre):
    """model.py:4-22 Encoder: Linear-ReLU x3 (ReLU after the last layer too)."""
    for i in (0, 2, 4):
        x = _relu(_linear(x, W[f"{pre}.model.{i}.weight"], W[f"{pre}.model.{i}.bias"]))
    return x


def model_forward_single(W, state, attrs, recv, send, group, action, phys, pstep, motion_clamp=100.0):
    """DynamicsPredictor.forward (model.py:130-342) for ONE batch element, index-list form.

    state (n_his,N,3); attrs (N,2); recv/send (E,) int; group (N,n_inst) = [p_instance ; 0] (model.py:264);
    action (N,3); phys (N,) with zeros for the n_s trailing tool particles (model.py:206-207).
    n_p (particles that get a prediction) = number of rows of p_instance = caller slices the output.
    Returns (pred_pos_all (N,3), motion_all (N,3)) - caller keeps the first n_p rows (model.py:335-338).
    """
    state = np.asarray(state, F32)
    n_his, N, _ = state.shape
    res = state[1:] - state[:-1]                                            # :156
    state_norm = np.concatenate([res, state[-1:]], 0)                       # :165
    snt = np.ascontiguousarray(state_norm.transpose(1, 0, 2)).reshape(N, n_his * 3)   # :166
    p_inputs = np.concatenate([attrs, phys[:, None], action], 1).astype(F32)  # :169,210,223 (state_dim=0)
    attrs_r, attrs_s = attrs[recv], attrs[send]                             # :253-254
    gdiff = np.abs(group[recv] - group[send]).sum(1, keepdims=True)         # :264-267
    pos_diff = snt[recv] - snt[send]                                        # :277-279
    rel_inputs = np.concatenate([attrs_r, attrs_s, gdiff, pos_diff], 1).astype(F32)  # :257,270,282
    p_enc = _encoder(p_inputs, W, "particle_encoder")                       # :297
    r_enc = _encoder(rel_inputs, W, "relation_encoder")                     # :303
    eff = p_enc
    Wrp, brp = W["relation_propagator.linear.weight"], W["relation_propagator.linear.bias"]
    Wpp, bpp = W["particle_propagator.linear.weight"], W["particle_propagator.linear.bias"]
    for _ in range(pstep):                                                  # :307
        x = np.concatenate([r_enc, eff[recv], eff[send]], 1)                # :312-318
        eff_rel = _relu(_linear(x, Wrp, brp))
        agg = np.zeros((N, eff_rel.shape[1]), F32)                          # :324
        np.add.at(agg, recv, eff_rel)
        eff = _relu(_linear(np.concatenate([p_enc, agg], 1), Wpp, bpp) + eff)   # :328-330 (res added before ReLU)
    h = _relu(_linear(eff, W["non_rigid_predictor.linear_0.weight"], W["non_rigid_predictor.linear_0.bias"]))
    h = _relu(_linear(h, W["non_rigid_predictor.linear_1.weight"], W["non_rigid_predictor.linear_1.bias"]))
    motion = _linear(h, W["non_rigid_predictor.linear_2.weight"], W["non_rigid_predictor.linear_2.bias"])  # :335
    pred = state[-1] + np.clip(motion, -F32(motion_clamp), F32(motion_clamp))   # :338
    return pred.astype(F32), motion.astype(F32)


def model_forward(W, state, attrs, edges, p_instance, action, physics_param, pstep):
    """Batched wrapper with the reference's argument meaning (model.py:130-131).

    state (B,n_his,N,3), attrs (B,N,2), edges = list of (recv, send), p_instance (B,n_p,n_inst),
    action (B,N,3), physics_param (B,1) or (B,n_p).  Returns pred_pos (B,n_p,3), pred_motion (B,n_p,3).
    """
    B, n_his, N, _ = state.shape
    n_p, n_inst = p_instance.shape[1], p_instance.shape[2]
    pos = np.zeros((B, n_p, 3), F32)
    mot = np.zeros((B, n_p, 3), F32)
    for b in range(B):
        phys = np.zeros(N, F32)
        pp = np.asarray(physics_param[b], F32).reshape(-1)
        phys[:n_p] = pp[0] if pp.size == 1 else pp                          # :191-207
        group = np.zeros((N, n_inst), F32)
        group[:n_p] = p_instance[b]                                         # :264
        p, m = model_forward_single(W, state[b], attrs[b], edges[b][0], edges[b][1], group, action[b], phys, pstep)
        pos[b], mot[b] = p[:n_p], m[:n_p]
    return pos, mot


# --------------------------------------------------------------------------- rollout driver
def decode_action(action, push_length):
    """plan_utils.py:11-20.  torch CPU cos/sin so the bits equal the reference's."""
    import torch
    a = torch.as_tensor(np.asarray(action, F32))
    x, z, th, ln = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    rep = ln.to(torch.int32)
    x_end = x - push_length * torch.cos(th)
    z_end = z - push_length * torch.sin(th)
    dec = torch.stack([x, z, x_end, z_end], -1)
    return dec.numpy(), rep.numpy()


def tool_keypoints(decoded, theta, task):
    """forward_dynamics.py:42-81 without the y row (y depends on the current particle cloud).

    decoded (...,4) [x0,z0,x1,z1], theta (...,) -> eef_xz (...,M,2), eef_delta (...,M,3) fp32.
    """
    import torch
    dec = torch.as_tensor(decoded)
    th = torch.as_tensor(np.asarray(theta, F32))
    pts = task["pusher_points"]
    ratio = task["sim_real_ratio"]
    lead = dec.shape[:-1]
    M = len(pts)
    if M not in (1, 5):
        raise NotImplementedError("pusher not implemented")                 # :77-78
    xz = torch.zeros(lead + (M, 2))
    delta = torch.zeros(lead + (M, 3))
    delta[..., 0] = (dec[..., 2] - dec[..., 0]).unsqueeze(-1)               # :48 / :56
    delta[..., 2] = (dec[..., 3] - dec[..., 1]).unsqueeze(-1)               # :50 / :58
    xz[..., 0, 0] = dec[..., 0]
    xz[..., 0, 1] = dec[..., 1]
    for k in range(1, M):                                                   # :65-75
        c = float(pts[k][1]) * ratio
        xz[..., k, 0] = dec[..., 0] + c * torch.sin(th)
        xz[..., k, 1] = dec[..., 1] - c * torch.cos(th)
    return xz.numpy(), delta.numpy()


def _rollout_candidate(W, pstep, obj0, obj_mask, eef_xz, eef_delta, repeat, task, y_mode, phys_val, max_nR,
                       trace=None):
    """One candidate, one look-ahead step: forward_dynamics.py:83-197 (and :278-393 for the masked variant).

    obj0 (N_o,3) start cloud (all n_his history frames equal it, :25/:38/:227); returns captured state or None.
    """
    N_o = obj0.shape[0]
    M = eef_xz.shape[0]
    N = N_o + M
    n_his = task["n_his"]
    grip = F32(0.01 * task["sim_real_ratio"]) if task["gripper_enable"] else None

    def tool_y(cloud):
        if y_mode == "min":
            y = cloud[:, 1].min()                                            # :40 / :163
        else:
            y = (cloud[:, 1] * obj_mask).sum(dtype=F32) / F32(obj_mask.sum())  # :235 / :359
        y = F32(y)
        return F32(y + grip) if grip is not None else y                     # :80-81 / :167-168

    eef = np.zeros((M, 3), F32)
    eef[:, 0], eef[:, 2] = eef_xz[:, 0], eef_xz[:, 1]
    eef[:, 1] = tool_y(obj0)
    cur = np.concatenate([obj0, eef], 0).astype(F32)
    hist = np.repeat(cur[None], n_his, 0)                                   # :83-85
    action = np.zeros((N, 3), F32)
    action[N_o:] = eef_delta                                                # :87-88
    attrs = np.zeros((N, 2), F32)
    attrs[:N_o, 0] = obj_mask.astype(F32)                                   # :92 / :287
    attrs[N_o:, 1] = 1                                                      # :93
    group = np.zeros((N, 1), F32)
    group[:int(obj_mask.sum()), 0] = 1                                      # :100-105 / :294-300 (first-count rows)
    mask = np.concatenate([obj_mask, np.ones(M, bool)])                     # :107-109 / :302-304
    tool = np.concatenate([np.zeros(N_o, bool), np.ones(M, bool)])          # :111-112
    phys = np.zeros(N, F32)
    phys[:N_o] = phys_val                                                   # :151 + model.py:197
    captured = None
    for ai in range(1, int(repeat) + 1):                                    # :156 (live steps only)
        recv, send = construct_edges_single(hist[-1], task["adj_thresh"], mask, tool, task["topk"],
                                            task["connect_tools_all"])      # :125 / :171
        if len(recv) > max_nR:
            raise Exception("Exceeds max dims")                             # utils.py:63-65 via :127-128
        pred, motion = model_forward_single(W, hist, attrs, recv, send, group, action, phys, pstep)
        pred = pred[:N_o]
        if trace is not None:
            trace.append({"recv": recv, "send": send, "state_last": hist[-1].copy(), "pred_pos": pred.copy()})
        if ai == repeat:
            captured = pred.copy()                                          # :160-161
        eef_cur = hist[-1, N_o:] + action[N_o:]                             # :164
        eef_cur[:, 1] = tool_y(pred)                                        # :163,166-168
        cur = np.concatenate([pred, eef_cur], 0).astype(F32)                # :170
        hist = np.concatenate([hist[1:], cur[None]], 0)                     # :176
    return captured


def dynamics(W, pstep, state, action, task, physics_param=0.5, trace=None):
    """forward_dynamics.py:12-205.  state (N_o,3); action (B,H,4).  Returns dict like the reference.
    physics_param: scalar, or (N_o,) per-particle values (the (B,n_p) branch of model.py:200-204).

    Candidates are independent, so each is stepped alone for exactly repeat[b,li] steps;
    the reference steps everyone to the batch max and discards the surplus (:156-161) - same outputs.
    `trace`, if a list, receives per-candidate lists of per-forward records.
    """
    state = np.asarray(state, F32)
    action = np.asarray(action, F32)
    B, H, _ = action.shape
    N_o = state.shape[0]
    dec, rep = decode_action(action, task["push_length"])                   # :23
    xz, delta = tool_keypoints(dec, action[..., 2], task)
    out = np.zeros((B, H, N_o, 3), F32)                                     # :32
    ones = np.ones(N_o, bool)
    for b in range(B):
        tr = [] if trace is not None else None
        obj = state
        for li in range(H):                                                 # :34
            if li > 0:
                obj = out[b, li - 1]                                        # :37-38
            cap = _rollout_candidate(W, pstep, obj, ones, xz[b, li], delta[b, li], rep[b, li], task, "min",
                                     np.asarray(physics_param, F32), task["max_nR"], tr)
            if cap is not None:
                out[b, li] = cap                                            # repeat==0 leaves zeros (:32,:160)
        if trace is not None:
            trace.append(tr)
    return {"state_seqs": out, "action_seqs": dec}


def dynamics_masked(W, pstep, state_init, state_mask, action, task, physics_param=0.5, trace=None):
    """forward_dynamics.py:209-399.  state_init (B,max_nobj,3), state_mask (B,max_nobj) bool, action (B,4)."""
    state_init = np.asarray(state_init, F32)
    state_mask = np.asarray(state_mask, bool)
    action = np.asarray(action, F32)
    B = state_init.shape[0]
    dec, rep = decode_action(action[:, None], task["push_length"])          # :218-223
    dec, rep = dec[:, 0], rep[:, 0]
    xz, delta = tool_keypoints(dec, action[:, 2], task)
    out = np.zeros_like(state_init)                                         # :233
    for b in range(B):
        tr = [] if trace is not None else None
        cap = _rollout_candidate(W, pstep, state_init[b], state_mask[b], xz[b], delta[b], rep[b], task, "mean",
                                 np.asarray(physics_param, F32), task["max_nR"], tr)
        if cap is not None:
            out[b] = cap
        if trace is not None:
            trace.append(tr)
    return {"state_seqs": out, "action_seqs": dec}


def rollout_work(W, pstep, state, action, task, physics_param=0.5):
    """Model forwards per candidate that remain once the forwards BEFORE its first tool contact are taken from one tool-free
    base rollout (the engine's contact-free prefix; checker for adaptigraph_amd.rollout_work / csrc/ag_graph.hip: k_contact_plan).
    Property of the reference it rests on: a tool takes part in an edge only if some object particle lies inside its radius -
    the radius test of graph.py:251-267 comes before top-k, and connect_tools_all's tool -> object edges are all-or-nothing on
    such a pair (:276-286) - so until then the object particles evolve as without a tool (forward_dynamics.py:156-176).
    -> (work (B,) int64, first (B,) int64): work = repeat - first + 1 for look-ahead step 0 (0 when it never touches) + the
    repeats of the later steps; first = 1-based forward of the first contact at look-ahead step 0, 0 = never."""
    state = np.asarray(state, F32)
    action = np.asarray(action, F32)
    B, H, _ = action.shape
    N_o = state.shape[0]
    dec, rep = decode_action(action, task["push_length"])
    xz, delta = tool_keypoints(dec, action[..., 2], task)
    M = xz.shape[2]
    R = int(max(0, rep[:, 0].max()))
    tr = []
    far = np.full((M, 2), 1.0e6, F32)                                       # a tool out of every particle's reach that stays there
    if R > 0:
        _rollout_candidate(W, pstep, state, np.ones(N_o, bool), far, np.zeros((M, 3), F32), R, task, "min",
                           np.asarray(physics_param, F32), 1 << 30, tr)
    S = [state] + [t["pred_pos"] for t in tr]                                # S_0 .. S_R
    grip = F32(0.01 * task["sim_real_ratio"]) if task["gripper_enable"] else None
    base_y = [F32(s[:, 1].min()) if grip is None else F32(F32(s[:, 1].min()) + grip) for s in S]    # forward_dynamics.py:40,163,167
    thr2 = F32(task["adj_thresh"]) * F32(task["adj_thresh"])                # graph.py:248-250
    work = np.zeros(B, np.int64)
    first = np.zeros(B, np.int64)
    for b in range(B):
        r0 = int(max(0, rep[b, 0]))
        tx, tz = xz[b, 0, :, 0].copy(), xz[b, 0, :, 1].copy()
        d = 0
        for ai in range(1, r0 + 1):                                         # graph of forward ai: objects S_(ai-1), tool after ai-1 advances
            P = S[ai - 1]
            for m in range(M):
                dx, dy, dz = P[:, 0] - tx[m], P[:, 1] - base_y[ai - 1], P[:, 2] - tz[m]
                dis = (dx * dx + dy * dy) + dz * dz                          # graph.py:251-252 (fp32, this order)
                if ((dis - thr2) < 0).any():                                # :267
                    d = ai
                    break
            if d:
                break
            tx, tz = (tx + delta[b, 0, :, 0]).astype(F32), (tz + delta[b, 0, :, 2]).astype(F32)   # forward_dynamics.py:164
        first[b] = d
        work[b] = (r0 - d + 1 if d else 0) + int(np.maximum(rep[b, 1:], 0).sum())
    return work, first


# --------------------------------------------------------------------------- helpers shared by tests / bench
# --------------------------------------------------------------------------- eval open-loop rollout step
def surface_bounds(obj_kp, ratio):
    """rollout.py:132-139 on numpy float32 scalars, as written there."""
    obj_kp = np.asarray(obj_kp, F32)
    max_y = np.max(obj_kp[:, 1]) * ratio
    min_y = np.min(obj_kp[:, 1])
    max_x = np.max(obj_kp[:, 0]) * ratio
    max_z = np.max(obj_kp[:, 2]) * ratio
    min_x = np.min(obj_kp[:, 0])
    min_x = (max_x - min_x) * (1 - ratio) + min_x
    min_z = np.min(obj_kp[:, 2])
    min_z = (max_z - min_z) * (1 - ratio) + min_z
    return dict(max_y=max_y, min_y=min_y, max_x=max_x, max_z=max_z, min_x=min_x, min_z=min_z)


def edges_with_backoff(pos, cfg, mask, tool_mask, bounds, trail=None):
    """construct_edges_from_states + the max_nR back-off of rollout.py:168-222: kNN down by knn_increment to min_kNN, then
    top-k down by one per attempt.  cfg: the dataset entries rollout.py:27-51 reads.  trail: (kNN, topk, n_rel) per attempt."""
    kw = dict(connect_tools_all=cfg["connect_tool_all"], connect_tools_surface=cfg["connect_tool_surface"],
              connect_tool_all_non_fixed=cfg["connect_tool_all_non_fixed"], **bounds)
    kNN, dec, k_now = cfg["knn_thresh"], cfg["topk"], cfg["topk"]
    while True:
        r, s = construct_edges_from_states(pos, cfg["adj_thresh"], mask, tool_mask, topk=k_now, kNN=kNN, **kw)
        if trail is not None:
            trail.append([float(kNN), int(k_now), len(r)])
        if len(r) <= cfg["max_nR"]:                                                       # :192-194 pad_torch fits
            return r, s
        if kNN <= cfg["min_kNN"]:                                                         # :199-211
            dec = dec - 1
            k_now = dec
        else:                                                                             # :212-222
            kNN = kNN - cfg["knn_increment"]
            k_now = cfg["topk"]


def eval_rollout_step(W, pstep, graph, eef_start, eef_end, cfg, trail=None):
    """One iteration of rollout_from_start_graph's loop (rollout.py:108-260) without its dataset side.  graph (numpy, one graph):
    'state' (n_his,N+M,3), 'action' (N+M,3), 'attrs' (N+M,2), 'edges' (recv, send), 'p_instance' (N,n_inst), 'physics' (N,),
    'obj_mask' (N,), 'state_mask', 'eef_mask' (N+M,).  -> (next graph, pred_state (N,3), pred_motion (N,3))."""
    n_p = graph["p_instance"].shape[0]
    pos, mot = model_forward(W, graph["state"][None], graph["attrs"][None], [graph["edges"]], graph["p_instance"][None],
                             graph["action"][None], graph["physics"][None], pstep)         # :112
    pred = pos[0]
    obj_kp = pred[graph["obj_mask"]]                                                      # :121
    bounds = surface_bounds(obj_kp, cfg["connect_tool_surface_ratio"])                    # :132-139
    states = np.concatenate([pred, np.asarray(eef_start, F32)], 0).astype(F32)            # :163
    delta = np.zeros_like(states)
    delta[n_p:n_p + len(eef_start)] = np.asarray(eef_end, F32) - np.asarray(eef_start, F32)   # :165-166
    edges = edges_with_backoff(states, cfg, graph["state_mask"], graph["eef_mask"], bounds, trail)
    hist = graph["state"]
    if cfg.get("store_rest_state"):
        hist = np.concatenate([hist[:1], hist[2:], states[None]], 0)                      # :224-229 the rest frame stays
    else:
        hist = np.concatenate([hist[1:], states[None]], 0)                                # :231-232
    nxt = dict(graph)
    nxt.update(state=hist.astype(F32), action=delta, edges=edges)
    return nxt, pred, mot[0]


# --------------------------------------------------------------------------- weights
def weights_from_npz(npz):
    return {k[3:]: np.asarray(npz[k], F32) for k in npz.files if k.startswith("w::")}


def random_weights(seed, nf=150, in_dim=6, rel_dim=17):
    """nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) init with a numpy RNG (synthetic benches)."""
    rng = np.random.default_rng(seed)
    shapes = {
        "particle_encoder.model.0": (nf, in_dim), "particle_encoder.model.2": (nf, nf),
        "particle_encoder.model.4": (nf, nf),
        "relation_encoder.model.0": (nf, rel_dim), "relation_encoder.model.2": (nf, nf),
        "relation_encoder.model.4": (nf, nf),
        "particle_propagator.linear": (nf, 2 * nf), "relation_propagator.linear": (nf, 3 * nf),
        "non_rigid_predictor.linear_0": (nf, nf), "non_rigid_predictor.linear_1": (nf, nf),
        "non_rigid_predictor.linear_2": (3, nf),
    }
    W = {}
    for k, (o, i) in shapes.items():
        bound = 1.0 / np.sqrt(i)
        W[k + ".weight"] = rng.uniform(-bound, bound, (o, i)).astype(F32)
        W[k + ".bias"] = rng.uniform(-bound, bound, (o,)).astype(F32)
    return W
