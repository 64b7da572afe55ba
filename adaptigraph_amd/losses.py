"""Cost functions of the planner with the reference's names and signatures (src/planning/losses.py:4-92), executing
on the HIP engine.  Inputs are torch tensors on the GPU; outputs are GPU tensors of the reference's shapes.

cloth_penalty normalises by a BATCH-GLOBAL maximum (losses.py:62).  When the candidate batch is sharded over ranks
pass `group=` (or have torch.distributed initialised and pass group=True for the default group): the maximum is then
all-reduced (MAX) so that every rank computes what the unsharded call would.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .context import default_engine, ptr, current_stream, _require_gpu

_KIND = {"rope": 0, "cloth": 1, "granular": 2}


def _global_max(t, group):
    m = t.max()
    if group is not None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            m = m.clone()
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=None if group is True else group)
    return m


def chamfer(x, y, x_mask=None, y_mask=None):
    """losses.py:4-10.  x (B,N,3), y (1|B,M,3) -> (B,).  Optional boolean masks select points (mean_chamfer)."""
    dev = _require_gpu(x.device)
    eng = default_engine(dev)
    x = x.to(torch.float32).contiguous()
    y = y.to(device=dev, dtype=torch.float32).contiguous()
    R, N, _ = x.shape
    By, M, _ = y.shape
    xm = x_mask.to(dev).to(torch.bool).contiguous().view(torch.uint8) if x_mask is not None else None
    ym = y_mask.to(dev).to(torch.bool).contiguous().view(torch.uint8) if y_mask is not None else None
    out = torch.empty(R, device=dev, dtype=torch.float32)
    eng.check(eng.lib.ag_cost_chamfer(eng.ctx, current_stream(dev), ptr(x), ptr(y), ptr(xm), ptr(ym), R, N, M, By, ptr(out)))
    return out


def mean_chamfer(state_pred, state_real, state_pred_mask, state_real_mask):
    """losses.py:12-24: per-pair masked chamfer, returned as a numpy float64 array like the reference."""
    out = chamfer(state_pred, state_real, state_pred_mask, state_real_mask)
    return out.detach().cpu().numpy().astype(np.float64)


def state_stats(state, box=None):
    """(R,N,3) -> (R,5) [box_loss, xmin, xmax, zmin, zmax] in one pass over the particles."""
    dev = _require_gpu(state.device)
    eng = default_engine(dev)
    state = state.to(torch.float32).contiguous()
    R, N, _ = state.shape
    out = torch.empty((R, 5), device=dev, dtype=torch.float32)
    box4 = None
    if box is not None:
        b = torch.as_tensor(box).detach().to("cpu", torch.float32)
        box4 = (C.c_float * 4)(float(b[0, 0]), float(b[0, 1]), float(b[1, 0]), float(b[1, 1]))
    eng.check(eng.lib.ag_cost_state_stats(eng.ctx, current_stream(dev), ptr(state), R, N, box4, ptr(out)))
    return out


def box_loss(state, target):
    """losses.py:26-35.  state (B,N,3), target (2,2) -> (B,)"""
    return state_stats(state, target)[:, 0].contiguous()


def _penalty_raw(kind, state_pred, action, state_init, sim_real_ratio):
    dev = _require_gpu(state_pred.device)
    eng = default_engine(dev)
    sp = state_pred.to(torch.float32).contiguous()
    act = action.to(device=dev, dtype=torch.float32).contiguous()
    si = state_init.to(device=dev, dtype=torch.float32).contiguous()
    B, H, N, _ = sp.shape
    assert act.shape[:2] == (B, H) and act.shape[2] >= 3 and si.shape == (N, 3)
    if act.shape[2] != 4:
        act = torch.cat([act, torch.zeros(B, H, 4 - act.shape[2], device=dev)], 2).contiguous()
    out = torch.empty((B, H, 2), device=dev, dtype=torch.float32)
    eng.check(eng.lib.ag_cost_penalty(eng.ctx, current_stream(dev), ptr(sp), ptr(act), ptr(si), B, H, N, _KIND[kind],
                                      float(sim_real_ratio), ptr(out)))
    return out


def rope_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:37-48 -> (B,H)"""
    return _penalty_raw("rope", state_pred, action, state_init, sim_real_ratio)[..., 0].contiguous()


def granular_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:66-92 -> (B,H)"""
    return _penalty_raw("granular", state_pred, action, state_init, sim_real_ratio)[..., 0].contiguous()


def cloth_penalty(state_pred, action, state_init, sim_real_ratio=10.0, group=None):
    """losses.py:50-64 -> (B,H).  `group`: all-reduce the batch-global maximum over the ranks sharing the batch."""
    raw = _penalty_raw("cloth", state_pred, action, state_init, sim_real_ratio)
    dev = raw.device
    eng = default_engine(dev)
    # :62-63 in one launch; a sharded batch all-reduces the maximum first and hands it in
    dmax = None if group is None else _global_max(raw[..., 1], group).reshape(1).to(dev, torch.float32).contiguous()
    out = torch.empty(raw.shape[:2], device=dev, dtype=torch.float32)
    eng.check(eng.lib.ag_cost_cloth_combine(eng.ctx, current_stream(dev), ptr(raw), ptr(dmax), raw.shape[0] * raw.shape[1], ptr(out)))
    return out
