"""adaptigraph_amd - MI355X-native GNN-dynamics rollout engine with AdaptiGraph's call signatures.

    from adaptigraph_amd import DynamicsPredictor, dynamics, dynamics_masked
    from adaptigraph_amd import construct_edges_from_states_batch, pad_torch, truncate_graph, decode_action

Hand-written HIP kernels (adaptigraph_amd/csrc) behind a C-ABI (include/adaptigraph_amd.h).  No CPU fallback.
"""
import os as _os


def _hw_queues_default():
    """HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
    serialise.  The engine runs a large batch on four in-library streams, the planner deals its chunk loop to six,
    torch.distributed's RCCL brings one more: eight queues unless the user chose otherwise (measured: 474.8 ->
    467.5 ms per bench step beside an RCCL group, planner call rope 165 -> 151 ms).  The HIP runtime reads the variable when it
    initialises, so a process that touched the GPU before this import keeps what it had: that case is reported (once, a warning)
    instead of passing silently, and `hw_queues` says which case applies - timings depend on it."""
    chosen = _os.environ.get("GPU_MAX_HW_QUEUES")
    info = {"GPU_MAX_HW_QUEUES": chosen, "set_by": "environment", "in_effect": None}
    if chosen is None:
        import torch as _torch
        late = _torch.cuda.is_initialized()
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"
        info.update(GPU_MAX_HW_QUEUES="8", set_by="adaptigraph_amd", in_effect=not late)
        if late:
            import warnings
            warnings.warn("adaptigraph_amd: HIP was initialised before the package was imported, so GPU_MAX_HW_QUEUES=8 cannot take "
                          "effect in this process (the runtime keeps its default of 4 hardware queues: the engine's four streams, the "
                          "planner's six and RCCL's then share queues and partly serialise). Import adaptigraph_amd - or export "
                          "GPU_MAX_HW_QUEUES=8 - before the first GPU call.", RuntimeWarning, stacklevel=3)
    return info


hw_queues = _hw_queues_default()

from .context import Engine, default_engine
from .forward_dynamics import dynamics, dynamics_masked, dynamics_mixed, rollout_work
from .graph import (EdgeList, construct_edges_from_states_batch, construct_edges_from_states, construct_edges_index,
                    construct_edges_with_backoff, pad_torch, truncate_graph)
from .model import DynamicsPredictor
from .plan_utils import decode_action
from .losses import chamfer, mean_chamfer, box_loss, rope_penalty, cloth_penalty, granular_penalty
from .costs import running_cost
from .physics_param_optimizer import dynamics_error, dynamics_error_sweep
from .mppi import angle_normalize, clip_actions, sample_action_seq, optimize_action_mppi, mpc_iteration
from .planner import Planner
from .rollout import rollout_eval, rollout_eval_step, surface_bounds

__all__ = ["hw_queues", "Engine", "default_engine", "dynamics", "dynamics_masked", "dynamics_mixed", "rollout_work", "EdgeList", "construct_edges_from_states_batch", "construct_edges_from_states",
           "construct_edges_index", "construct_edges_with_backoff", "pad_torch", "truncate_graph", "DynamicsPredictor", "decode_action", "chamfer",
           "mean_chamfer", "box_loss", "rope_penalty", "cloth_penalty", "granular_penalty", "running_cost", "dynamics_error", "dynamics_error_sweep", "angle_normalize",
           "clip_actions", "sample_action_seq", "optimize_action_mppi", "mpc_iteration", "Planner", "rollout_eval", "rollout_eval_step", "surface_bounds"]
