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


def construct_edges_from_states(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, max_y=None,
                                min_y=None, max_x=None, max_z=None, min_x=None, min_z=None, connect_tools_surface=False,
                                connect_tool_all_non_fixed=True, kNN=1.0, as_index=False):
    """Drop-in for the single-graph builder (graph.py:68-231), default-argument path: states (N,3), mask/tool_mask (N,)
    -> dense one-hot (Rr, Rs) of shape (n_rel, N) (or an EdgeList with as_index=True).  The tool-surface / kNN /
    non-fixed-particle branches (graph.py:125-221) need max_y etc.; they are not implemented."""
    if (connect_tool_all_non_fixed and max_y is not None and min_y is not None) or \
            (connect_tools_surface and None not in (max_y, max_x, min_x, max_z, min_z)):
        raise NotImplementedError("tool-surface / non-fixed-particle edge rules (graph.py:125-221) are not implemented")
    import numpy as np
    dev = _require_gpu(states.device)
    eng = default_engine(dev)
    pos = states.to(torch.float32).contiguous()
    N = pos.shape[0]
    thr = float(adj_thresh)
    thr2 = float(np.float32(thr * thr))                                     # double product, one fp32 rounding (:86,101)
    cull = float(np.nextafter(np.float32(abs(thr)), np.float32(np.inf)))    # cull^2 >= thr2 whatever the rounding did
    mask_u8 = mask.to(dev).to(torch.bool).contiguous().view(torch.uint8)
    tool_u8 = tool_mask.to(dev).to(torch.bool).contiguous().view(torch.uint8)
    k = min(N, int(topk))
    m = int(tool_mask.to(torch.bool).sum().item())
    edge_cap = max(1, N * (k + m) if k < N else N * N)
    recv = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    send = torch.empty((1, edge_cap), device=dev, dtype=torch.int32)
    row_ptr = torch.empty((1, N + 1), device=dev, dtype=torch.int32)
    n_edges = torch.empty((1,), device=dev, dtype=torch.int32)
    eng.check(eng.lib.ag_build_edges_single(eng.ctx, current_stream(dev), ptr(pos), ptr(mask_u8), ptr(tool_u8), N, thr2,
                                            cull, int(topk), int(bool(connect_tools_all)), edge_cap, ptr(recv), ptr(send),
                                            ptr(row_ptr), ptr(n_edges)))
    el = EdgeList(recv, send, row_ptr, n_edges, N)
    if as_index:
        return el
    Rr, Rs = el.to_dense()
    return Rr[0], Rs[0]


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
