"""Engine context: owns the C-side ag_ctx (weights + workspace) for one (process, device)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

STATE_DICT_ORDER = [
    "particle_encoder.model.0", "particle_encoder.model.2", "particle_encoder.model.4",
    "relation_encoder.model.0", "relation_encoder.model.2", "relation_encoder.model.4",
    "particle_propagator.linear", "relation_propagator.linear",
    "non_rigid_predictor.linear_0", "non_rigid_predictor.linear_1", "non_rigid_predictor.linear_2",
]


def _require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("adaptigraph_amd runs on an AMD GPU (torch device 'cuda:N' under ROCm); "
                           f"got device={device}. There is no CPU fallback.")
    return device


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """One ag_ctx.  `pstep` etc. = model_config of the reference (src/dynamics/gnn/model.py:78-123)."""

    def __init__(self, device, pstep=3, nf=150, n_his=4, in_dim=6, rel_dim=17, motion_clamp=100.0):
        self.device = _require_gpu(device)
        self.lib = _lib.load()
        self._ctx = C.c_void_p(0)
        dims = _lib.AgDims(nf, n_his, pstep, in_dim, rel_dim, motion_clamp)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.ag_ctx_create(idx, C.byref(dims), C.byref(self._ctx))
        if rc != 0:
            msg = self.lib.ag_last_error(self._ctx).decode() if self._ctx else "ag_ctx_create failed"
            if self._ctx:
                self.lib.ag_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p(0)
            if rc == _lib.AG_ERR_UNSUPPORTED:
                raise NotImplementedError(msg)
            raise RuntimeError(msg)
        self.pstep = pstep
        self._weights_key = None
        self.precision = "fp32"
        import os
        if os.environ.get("AG_PRECISION", "fp32") != "fp32":
            self.set_precision(os.environ["AG_PRECISION"])

    def close(self):
        if self._ctx:
            self.lib.ag_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p(0)

    def __del__(self):  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- error mapping: same exception types/messages the reference raises on this path
    def check(self, rc):
        if rc == 0:
            return
        msg = self.lib.ag_last_error(self._ctx).decode()
        if rc == _lib.AG_ERR_MAX_NR:
            raise Exception("Exceeds max dims")          # src/dynamics/utils.py:63-65 raises a bare Exception
        if rc == _lib.AG_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == _lib.AG_ERR_INVALID:
            raise AssertionError(msg)                    # the reference asserts on shapes (model.py:187,222,240,...)
        raise RuntimeError(f"adaptigraph_amd: {msg} (code {rc})")

    def load_state_dict_tensors(self, sd):
        """sd: mapping with the 22 reference keys."""
        host = []
        for base in STATE_DICT_ORDER:
            for suffix in (".weight", ".bias"):
                host.append(sd[base + suffix].detach().to("cpu", torch.float32).contiguous())
        arr = (C.c_void_p * len(host))(*[t.data_ptr() for t in host])
        self.check(self.lib.ag_ctx_load_weights(self._ctx, arr, len(host)))

    def set_precision(self, mode):
        """'fp32' (default, exact v_mfma_f32) or 'bf16x3' (3-way bf16 split on the bf16 matrix pipe, fp32-grade accuracy)."""
        self.check(self.lib.ag_ctx_set_precision(self._ctx, {"fp32": 0, "bf16x3": 1}[mode]))
        self.precision = mode

    def set_option(self, name, value):
        """Per-context switch between bit-identical execution paths (include/adaptigraph_amd.h: ag_ctx_set_option)."""
        self.check(self.lib.ag_ctx_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int32(0)
        self.check(self.lib.ag_ctx_get_option(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """Context manager: set the given options, restore the previous values on exit."""
        eng = self

        class _Scope:
            def __enter__(self):
                self.old = {k: eng.get_option(k) for k in kw}
                for k, v in kw.items():
                    eng.set_option(k, v)
                return eng

            def __exit__(self, *a):
                for k, v in self.old.items():
                    eng.set_option(k, v)
        return _Scope()

    def rollout_counts(self):
        """(executed, needed) candidate-forwards of the last rollout call (ag_ctx_rollout_counts)."""
        ex, need = C.c_int64(0), C.c_int64(0)
        self.check(self.lib.ag_ctx_rollout_counts(self._ctx, C.byref(ex), C.byref(need)))
        return ex.value, need.value

    def launch_counts(self):
        """(model forwards per launch chunk enqueued by the last rollout call, what the loop bounds alone give): ag_ctx_launch_counts."""
        out = (C.c_int64 * 2)()
        self.check(self.lib.ag_ctx_launch_counts(self._ctx, out))
        return int(out[0]), int(out[1])

    def alloc_counts(self):
        """device / pinned allocations and frees, event and stream creations this context has made so far (ag_ctx_alloc_counts)."""
        out = (C.c_int64 * 1)()
        self.check(self.lib.ag_ctx_alloc_counts(self._ctx, out))
        return int(out[0])

    def share_counts(self):
        """Shared first forward of the last rollout call (ag_ctx_share_counts): (edges of the once-per-call base encode,
        edge slots served by the shared table, edge slots the candidates encoded themselves at that forward)."""
        out = (C.c_int64 * 3)()
        self.check(self.lib.ag_ctx_share_counts(self._ctx, out))
        return int(out[0]), int(out[1]), int(out[2])

    def set_chunk(self, n):
        self.check(self.lib.ag_ctx_set_chunk(self._ctx, int(n)))

    def set_profiling(self, families, keep_streams=False):
        """HIP-event timing of the listed kernel families.  Pins the rollout to one stream unless keep_streams."""
        mask = 0
        for f in families:
            mask |= 1 << _lib.KERNEL_FAMILIES.index(f)
        if keep_streams and mask:
            mask |= 1 << 30
        self.check(self.lib.ag_ctx_set_profiling(self._ctx, mask))

    def kernel_stats(self, family):
        ms, n = C.c_double(0), C.c_int64(0)
        self.check(self.lib.ag_ctx_kernel_stats(self._ctx, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_stats(self):
        self.check(self.lib.ag_ctx_reset_stats(self._ctx))

    @property
    def ctx(self):
        return self._ctx


_SIDE_STREAMS = {}


def side_streams(device, n):
    """The same n side streams for every dealt evaluation on a device (dynamics_error_sweep, dynamics_mixed): the engine keeps one
    call slot per caller stream, so fresh streams per call would only make it recycle slots."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device))
    return have[:n]


_default_engines = {}


def default_engine(device):
    """Weight-less engine used by the stand-alone graph functions."""
    device = _require_gpu(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _default_engines:
        _default_engines[key] = Engine(device)
    return _default_engines[key]
