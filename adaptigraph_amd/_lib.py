"""ctypes binding of the C-ABI in include/adaptigraph_amd.h.

The product path has NO CPU fallback: if the HIP library is missing this module raises at import of the symbols,
and every op needs a ROCm device.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADAPTIGRAPH_AMD_LIB: load another build of the same C-ABI instead (the diagnostic build libadaptigraph_hip_diag.so, for
# tools/ and the injected-failure test).  The product never sets it.
LIB_PATH = os.environ.get("ADAPTIGRAPH_AMD_LIB") or os.path.join(_HERE, "csrc", "libadaptigraph_hip.so")

AG_OK = 0
AG_ERR_INVALID = -1
AG_ERR_HIP = -2
AG_ERR_MAX_NR = -3
AG_ERR_UNSUPPORTED = -4
AG_ERR_NO_WEIGHTS = -5

KERNEL_FAMILIES = ["edge_count", "edge_emit", "prep", "node_enc", "edge_enc", "mp", "node_prop", "node_final",
                   "roll_init", "roll_update", "cost"]

# exactly the symbols include/adaptigraph_amd.h declares (tests/test_abi.py checks both directions)
EXPORTS = ["ag_abi_version", "ag_ctx_create", "ag_ctx_destroy", "ag_last_error", "ag_ctx_load_weights",
           "ag_ctx_set_chunk", "ag_build_edges", "ag_forward", "ag_rollout", "ag_rollout_async",
           "ag_ctx_set_profiling", "ag_ctx_kernel_stats", "ag_ctx_reset_stats",
           "ag_cost_chamfer", "ag_cost_state_stats", "ag_cost_penalty", "ag_ctx_set_precision", "ag_build_edges_single",
           "ag_edges_apply_tool_rule", "ag_mppi_sample", "ag_mppi_update", "ag_mppi_clip",
           "ag_ctx_set_option", "ag_ctx_get_option", "ag_ctx_rollout_counts", "ag_rollout_actions", "ag_ctx_share_counts", "ag_ctx_launch_counts", "ag_cost_reward", "ag_cost_cloth_combine",
           "ag_ctx_alloc_counts", "ag_rollout_work"]

OPTIONS = ["streams", "chunk", "latency", "ragged", "ell_graph", "self_dedupe", "repeat_sort", "edge_wgs", "edge_block_min",
           "enc_persist", "stagger_us", "device_decode", "zigzag", "share_first", "share_prefix", "stream_min_rows", "pipeline_fork"]


class AgDims(C.Structure):
    _fields_ = [("nf", C.c_int32), ("n_his", C.c_int32), ("pstep", C.c_int32), ("in_dim", C.c_int32),
                ("rel_dim", C.c_int32), ("motion_clamp", C.c_float)]


class AgRolloutParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("N_o", C.c_int32), ("M", C.c_int32), ("topk", C.c_int32),
                ("connect_tools_all", C.c_int32), ("max_nR", C.c_int32), ("y_mode", C.c_int32),
                ("adj_thresh", C.c_float), ("gripper_offset", C.c_float), ("gripper_enable", C.c_int32),
                ("physics_param", C.c_float)]


_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"adaptigraph_amd: HIP library not built ({LIB_PATH}). Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or adaptigraph_amd/csrc/build.sh. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.ag_abi_version.restype = C.c_uint32
    lib.ag_abi_version.argtypes = []
    lib.ag_ctx_create.argtypes = [i32, C.POINTER(AgDims), C.POINTER(vp)]
    lib.ag_ctx_destroy.argtypes = [vp]
    lib.ag_last_error.argtypes = [vp]
    lib.ag_last_error.restype = C.c_char_p
    lib.ag_ctx_load_weights.argtypes = [vp, C.POINTER(vp), i32]
    lib.ag_ctx_set_chunk.argtypes = [vp, i32]
    lib.ag_ctx_set_precision.argtypes = [vp, i32]
    lib.ag_ctx_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.ag_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i32)]
    lib.ag_ctx_rollout_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ag_ctx_share_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ag_ctx_launch_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ag_ctx_alloc_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ag_build_edges.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.ag_build_edges_single.argtypes = [vp, vp, vp, vp, vp, i32, f32, f32, i32, i32, i32, vp, vp, vp, vp]
    lib.ag_edges_apply_tool_rule.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, C.c_double, i32, vp, vp, vp, vp]
    lib.ag_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.ag_rollout.argtypes = [vp, vp, C.POINTER(AgRolloutParams), vp, vp, vp, vp, vp, vp, vp]  # ..., h_repeat, d_phys_vec, d_state_seqs
    lib.ag_rollout_async.argtypes = [vp, vp, C.POINTER(AgRolloutParams), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ag_rollout_actions.argtypes = [vp, vp, C.POINTER(AgRolloutParams), vp, vp, f32, vp, i32, vp, vp, vp, vp]
    lib.ag_rollout_work.argtypes = [vp, vp, C.POINTER(AgRolloutParams), vp, vp, f32, vp, i32, vp, vp]
    lib.ag_cost_chamfer.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.ag_cost_state_stats.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.ag_cost_penalty.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.ag_cost_reward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.ag_cost_cloth_combine.argtypes = [vp, vp, vp, vp, C.c_int64, vp]
    lib.ag_mppi_sample.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]
    lib.ag_mppi_update.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp]
    lib.ag_mppi_clip.argtypes = [vp, vp, vp, vp, vp, C.c_int64, vp]
    lib.ag_ctx_set_profiling.argtypes = [vp, i32]
    lib.ag_ctx_kernel_stats.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ag_ctx_reset_stats.argtypes = [vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("ag_abi_version", "ag_last_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib
