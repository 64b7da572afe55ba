"""adaptigraph_amd - MI355X-native GNN-dynamics rollout engine with AdaptiGraph's call signatures.

    from adaptigraph_amd import DynamicsPredictor, dynamics, dynamics_masked
    from adaptigraph_amd import construct_edges_from_states_batch, pad_torch, truncate_graph, decode_action

Hand-written HIP kernels (adaptigraph_amd/csrc) behind a C-ABI (include/adaptigraph_amd.h).  No CPU fallback.
"""
import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.
# The engine runs a large batch on four in-library streams, the planner deals its chunk loop to six, torch.distributed's RCCL
# brings one more: eight queues unless the user chose otherwise.  Read by the HIP runtime when it initialises - a process that
# has already touched the GPU keeps what it had (measured: 474.8 -> 467.5 ms per bench step beside an RCCL
# group, planner call rope 165 -> 151 ms).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
from .rollout import rollout_eval_step, surface_bounds

__all__ = ["Engine", "default_engine", "dynamics", "dynamics_masked", "dynamics_mixed", "rollout_work", "EdgeList", "construct_edges_from_states_batch", "construct_edges_from_states",
           "construct_edges_index", "construct_edges_with_backoff", "pad_torch", "truncate_graph", "DynamicsPredictor", "decode_action", "chamfer",
           "mean_chamfer", "box_loss", "rope_penalty", "cloth_penalty", "granular_penalty", "running_cost", "dynamics_error", "dynamics_error_sweep", "angle_normalize",
           "clip_actions", "sample_action_seq", "optimize_action_mppi", "mpc_iteration", "Planner", "rollout_eval_step", "surface_bounds"]
