"""Graph construction on the GPU.  Mirrors reference src/dynamics/dataset/graph.py:233-298
(construct_edges_from_states_batch) and src/dynamics/utils.py:49-69,150-160 (pad_torch, truncate_graph).

The native representation is an index-list graph (EdgeList: recv/send/row_ptr, CSR by receiver).  The dense one-hot
(Rr, Rs) pair the reference returns is produced on request for callers that still want it.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .context import default_engine, ptr, current_stream, _require_gpu


@dataclass
class EdgeList:
    """Edges of a batch of graphs, per batch element sorted by (receiver, sender) = the reference's nonzero order."""
    recv: torch.Tensor      # (B, edge_cap) int32
    send: torch.Tensor      # (B, edge_cap) int32
    row_ptr: torch.Tensor   # (B, N+1) int32 CSR offsets by receiver
    n_edges: torch.Tensor   # (B,) int32
    N: int

    @property
    def edge_cap(self):
        return self.recv.shape[1]

    def to_dense(self, n_rel=None):
        """(Rr, Rs) exactly as graph.py:288-298 builds them: (B, max_b n_edges, N) one-hot fp32, zero-padded rows."""
        B = self.recv.shape[0]
        n_rel = int(self.n_edges.max().item()) if n_rel is None else n_rel
        dev = self.recv.device
        Rr = torch.zeros((B, n_rel, self.N), device=dev, dtype=torch.float32)
        Rs = torch.zeros((B, n_rel, self.N), device=dev, dtype=torch.float32)
        e = torch.arange(n_rel, device=dev)[None, :].expand(B, n_rel)
        valid = e < self.n_edges[:, None]
        b_idx = torch.arange(B, device=dev)[:, None].expand(B, n_rel)[valid]
        e_idx = e[valid]
        Rr[b_idx, e_idx, self.recv[:, :n_rel][valid].long()] = 1
        Rs[b_idx, e_idx, self.send[:, :n_rel][valid].long()] = 1
        return Rr, Rs

    @staticmethod
    def from_dense(Rr, Rs):
        """Dense one-hot (B,E,N) -> index lists.  Zero rows are padding (the convention of pad_torch/truncate_graph and
        of the reference's viz, src/dynamics/rollout/graph.py:215-217).  Rows are re-sorted by receiver (stable) when
        the caller's order is not CSR; this only changes the summation order of the scatter (model.py:324)."""
        B, E, N = Rr.shape
        valid = Rr.sum(-1) > 0
        recv = Rr.argmax(-1).to(torch.int32)
        send = Rs.argmax(-1).to(torch.int32)
        key = torch.where(valid, recv.long(), torch.full_like(recv, N, dtype=torch.long))
        order = torch.argsort(key, dim=1, stable=True)
        recv = torch.gather(recv, 1, order).contiguous()
        send = torch.gather(send, 1, order).contiguous()
        n_edges = valid.sum(1).to(torch.int32)
        key_sorted = torch.gather(key, 1, order)
        counts = torch.zeros((B, N + 1), device=Rr.device, dtype=torch.long)
        counts.scatter_add_(1, key_sorted, torch.ones_like(key_sorted))
        row_ptr = torch.zeros((B, N + 1), device=Rr.device, dtype=torch.int32)
        row_ptr[:, 1:] = torch.cumsum(counts[:, :N], 1).to(torch.int32)
        if E == 0:
            recv = torch.zeros((B, 1), device=Rr.device, dtype=torch.int32)
            send = torch.zeros((B, 1), device=Rr.device, dtype=torch.int32)
        return EdgeList(recv, send, row_ptr.contiguous(), n_edges.contiguous(), N)


def construct_edges_index(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, edge_cap=None,
                          engine=None):
    """Index-list form of construct_edges_from_states_batch.  Same arguments as graph.py:233; returns EdgeList.

    edge_cap: capacity per batch element (default: the structural bound N*(min(topk,N)+M)).  If a graph has more edges
    than edge_cap nothing is written for it and n_edges still reports the true count.
    """
    dev = _require_gpu(states.device)
    eng = engine or default_engine(dev)
    states = states.to(torch.float32).contiguous()
    B, N, sd = states.shape
    assert sd == 3, "state_dim must be 3"
    mask_u8 = mask.to(torch.bool).contiguous().view(torch.uint8)
    tool_u8 = tool_mask.to(torch.bool).contiguous().view(torch.uint8)
    thr_vec = None
    if isinstance(adj_thresh, torch.Tensor):
        thr_vec = adj_thresh.to(device=dev, dtype=torch.float32).contiguous()
        assert thr_vec.numel() == B
        thr = 0.0
    else:
        thr = float(adj_thresh)
    if edge_cap is None:
        k = min(N, int(topk))
        m = int(tool_mask.to(torch.bool).sum(1).max().item())
        edge_cap = N * (k + m) if k < N else N * N
    edge_cap = max(1, int(edge_cap))
    recv = torch.empty((B, edge_cap), device=dev, dtype=torch.int32)
    send = torch.empty((B, edge_cap), device=dev, dtype=torch.int32)
    row_ptr = torch.empty((B, N + 1), device=dev, dtype=torch.int32)
    n_edges = torch.empty((B,), device=dev, dtype=torch.int32)
    eng.check(eng.lib.ag_build_edges(eng.ctx, current_stream(dev), ptr(states), ptr(mask_u8), ptr(tool_u8), B, N, thr,
                                     ptr(thr_vec), int(topk), int(bool(connect_tools_all)), edge_cap, ptr(recv),
                                     ptr(send), ptr(row_ptr), ptr(n_edges)))
    return EdgeList(recv, send, row_ptr, n_edges, N)


def construct_edges_from_states_batch(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False):
    """Drop-in for graph.py:233-298: returns dense one-hot (Rr, Rs) of shape (B, n_rel, N)."""
    el = construct_edges_index(states, adj_thresh, mask, tool_mask, topk, connect_tools_all)
    if int((el.n_edges > el.edge_cap).any().item()):
        raise RuntimeError("internal: structural edge bound exceeded")
    return el.to_dense()


_PLANES = ["max_y", "min_x", "max_x", "min_z", "max_z"]            # graph.py:38 order


def _plane_side(name, pos, max_y, max_x, max_z, min_x, min_z):
    """graph.py:45-66 on particle vectors: which particles lie beyond the named bounding plane (torch's own
    tensor-vs-scalar comparison, so the scalar is rounded exactly as in the reference)."""
    if name == "max_y":
        return pos[:, 1] >= max_y
    if name == "max_x":
        return pos[:, 0] >= max_x
    if name == "max_z":
        return pos[:, 2] >= max_z
    if name == "min_x":
        return pos[:, 0] <= min_x
    if name == "min_z":
        return pos[:, 2] <= min_z
    raise Exception("Unknown plane for connecting tool to surface object particles!!")


def _apply_tool_rule(eng, dev, el, pos, mask_u8, tool_u8, n_tools, subset, kNN):
    """ag_edges_apply_tool_rule on a single-graph EdgeList -> new EdgeList."""
    N = el.N
    edge_cap = max(1, min(el.edge_cap, N * N) + N * n_tools)                # every rule adds at most N*M tool edges (no read-back)
    recv = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    send = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    row_ptr = torch.empty((1, N + 1), device=dev, dtype=torch.int32)
    n_edges = torch.empty((1,), device=dev, dtype=torch.int32)
    sub_u8 = subset.to(dev).to(torch.bool).contiguous().view(torch.uint8)
    eng.check(eng.lib.ag_edges_apply_tool_rule(eng.ctx, current_stream(dev), ptr(pos), ptr(mask_u8), ptr(tool_u8), N, n_tools,
                                               ptr(el.send), ptr(el.row_ptr), ptr(sub_u8), float(kNN), edge_cap, ptr(recv),
                                               ptr(send), ptr(row_ptr), ptr(n_edges)))
    return EdgeList(recv, send, row_ptr, n_edges, N)        # (n_edges < 0 = tool count mismatch: checked where n_edges is next read)


def _tool_sender_edges(el, tool_b):
    """adj[obj_tool_mask_2].sum() (graph.py:128-129, :178-179): edges whose sender is a tool particle.  (Receivers of
    edges are valid particles by construction, which is the other half of obj_tool_mask_2.)"""
    live = torch.arange(el.edge_cap, device=el.send.device) < el.n_edges[0]          # one read-back instead of two
    return int((tool_b[el.send[0].clamp(0, el.N - 1).long()] & live).sum().item())


def construct_edges_from_states(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, max_y=None,
                                min_y=None, max_x=None, max_z=None, min_x=None, min_z=None, connect_tools_surface=False,
                                connect_tool_all_non_fixed=True, kNN=1.0, as_index=False):
    """Drop-in for the single-graph builder (graph.py:68-231): states (N,3), mask/tool_mask (N,) -> dense one-hot
    (Rr, Rs) of shape (n_rel, N) (or an EdgeList with as_index=True).

    The radius / top-k / connect_tools_all part runs in ag_build_edges_single.  The two optional tool rules
    (graph.py:125-171 'tool to all non-fixed particles' with its flat kNN filter, :173-221 'tool to the two closest
    surface planes') are scalar decisions here - the same Python expressions as the reference, so thresholds round
    the same way - followed by ag_edges_apply_tool_rule on the device.  Like the reference this path synchronises
    (graph.py:129, :223)."""
    import numpy as np
    dev = _require_gpu(states.device)
    eng = default_engine(dev)
    pos = states.to(torch.float32).contiguous()
    N = pos.shape[0]
    thr = float(adj_thresh)
    thr2 = float(np.float32(thr * thr))                                     # double product, one fp32 rounding (:86,101)
    cull = float(np.nextafter(np.float32(abs(thr)), np.float32(np.inf)))    # cull^2 >= thr2 whatever the rounding did
    mask_b = mask.to(dev).to(torch.bool).contiguous()
    tool_b = tool_mask.to(dev).to(torch.bool).contiguous()
    mask_u8, tool_u8 = mask_b.view(torch.uint8), tool_b.view(torch.uint8)
    k = min(N, int(topk))
    m = int(tool_b.sum().item())
    edge_cap = max(1, N * (k + m) if k < N else N * N)
    recv = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    send = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    row_ptr = torch.empty((1, N + 1), device=dev, dtype=torch.int32)
    n_edges = torch.empty((1,), device=dev, dtype=torch.int32)
    eng.check(eng.lib.ag_build_edges_single(eng.ctx, current_stream(dev), ptr(pos), ptr(mask_u8), ptr(tool_u8), N, thr2,
                                            cull, int(topk), int(bool(connect_tools_all)), edge_cap, ptr(recv), ptr(send),
                                            ptr(row_ptr), ptr(n_edges)))
    el = EdgeList(recv, send, row_ptr, n_edges, N)

    if connect_tool_all_non_fixed and max_y is not None and min_y is not None:          # graph.py:125
        check = _tool_sender_edges(el, tool_b)                                          # :128-129
        threshold = (max_y - min_y) * 0.1 + min_y                                       # :134 bottom 10 % is fixed
        if check > 0:
            subset = (pos[:, 1] > threshold) & mask_b                                   # :138-143
            el = _apply_tool_rule(eng, dev, el, pos, mask_u8, tool_u8, m, subset, kNN)  # :144-170

    if connect_tools_surface and max_y is not None and max_x is not None and min_x is not None and max_z is not None \
            and min_z is not None:                                                      # graph.py:173
        check = _tool_sender_edges(el, tool_b)                                          # :178-179
        if check > 0:
            # :190-194 index s_receiv with the 0/1 VALUES of adj[obj_tool_mask_2]: each of its entries selects particle
            # 0 or particle 1, broadcast over N senders.  Reproduced as written: n0 zeros, n1 ones.
            n1 = check
            n0 = int(mask_b.sum().item()) * m - check
            p01 = pos[:2].cpu()
            def plane_dist(axis, bound):
                d = (p01[:, axis] - bound) ** 2                                         # fp32, like the reference
                return N * (n0 * float(d[0]) + n1 * float(d[min(1, N - 1)]))
            values = [plane_dist(1, max_y), plane_dist(0, min_x), plane_dist(0, max_x), plane_dist(2, min_z),
                      plane_dist(2, max_z)]                                             # :36-37
            order = np.argsort(values)                                                  # :39
            first, second = _PLANES[order[0]], _PLANES[order[1]]
            subset = _plane_side(first, pos, max_y, max_x, max_z, min_x, min_z) & \
                _plane_side(second, pos, max_y, max_x, max_z, min_x, min_z) & mask_b     # :197-207
            el = _apply_tool_rule(eng, dev, el, pos, mask_u8, tool_u8, m, subset, 1.0)  # :208-218

    if as_index:
        return el
    if int(el.n_edges[0].item()) < 0:
        raise RuntimeError("internal: tool count passed to ag_edges_apply_tool_rule does not match tool_mask")
    Rr, Rs = el.to_dense()
    return Rr[0], Rs[0]


def construct_edges_with_backoff(states, adj_thresh, mask, tool_mask, topk, max_nR, knn_thresh=1.0, min_kNN=1.0,
                                 knn_increment=0.1, as_index=False, trail=None, **rules):
    """The max_nR back-off loop the reference repeats around construct_edges_from_states (rollout.py:173-222,
    rollout/graph.py:508-543, dataset.py:310-350): if the graph does not fit max_nR, first shrink the tool's kNN
    fraction by knn_increment down to min_kNN, then lower top-k by one per attempt.  `rules` are the remaining keyword
    arguments of construct_edges_from_states (connect_tools_all, max_y, ...).  Returns (Rr, Rs) padded to max_nR, or with
    as_index the EdgeList of the graph that fitted (the fit test is then one integer read per attempt instead of the dense
    pad_torch).  trail (list): receives (kNN, topk, n_rel) of every attempt."""
    kNN = knn_thresh
    decrease_topK = topk
    k_now = topk
    while True:
        el = construct_edges_from_states(states, adj_thresh, mask, tool_mask, topk=k_now, kNN=kNN, as_index=True, **rules)
        n_rel = int(el.n_edges[0].item())
        if n_rel < 0:
            raise RuntimeError("internal: tool count passed to ag_edges_apply_tool_rule does not match tool_mask")
        if trail is not None:
            trail.append((float(kNN), int(k_now), n_rel))
        if n_rel <= max_nR:                                                             # rollout.py:192-194 pad_torch fits
            if as_index:
                return el
            Rr, Rs = el.to_dense(n_rel)
            return pad_torch(Rr[0], max_nR), pad_torch(Rs[0], max_nR)
        if kNN <= min_kNN:                                                              # rollout.py:199-211
            decrease_topK = decrease_topK - 1
            if decrease_topK < 1:
                raise Exception("Exceeds max dims")                                     # (the reference would loop on: utils.py:63-65)
            k_now = decrease_topK
        else:                                                                           # rollout.py:212-222
            kNN = kNN - knn_increment
            k_now = topk


def pad_torch(x, max_dim, dim=0):
    """src/dynamics/utils.py:49-69: zero-pad `dim` to max_dim, raise Exception('Exceeds max dims') when larger."""
    if dim == 0:
        x_dim = x.shape[0]
        out = torch.zeros((max_dim, x.shape[1]), dtype=x.dtype, device=x.device)
        if x_dim > max_dim:
            raise Exception("Exceeds max dims")
        out[:x_dim] = x
    elif dim == 1:
        x_dim = x.shape[1]
        out = torch.zeros((x.shape[0], max_dim, x.shape[2]), dtype=x.dtype, device=x.device)
        if x_dim > max_dim:
            raise Exception("Exceeds max dims")
        out[:, :x_dim] = x
    else:
        raise ValueError("pad_torch supports dim 0 or 1")
    return out


def truncate_graph(data):
    """src/dynamics/utils.py:150-160: cut Rr/Rs back to the largest non-zero row count in the batch."""
    Rr, Rs = data["Rr"], data["Rs"]
    n_Rr = int((Rr.sum(-1) > 0).sum(1).max().item())
    n_Rs = int((Rs.sum(-1) > 0).sum(1).max().item())
    n = max(n_Rr, n_Rs)
    data["Rr"] = Rr[:, :n, :]
    data["Rs"] = Rs[:, :n, :]
    return data
