"""adaptigraph_amd - MI355X-native GNN-dynamics rollout engine with AdaptiGraph's call signatures.

    from adaptigraph_amd import DynamicsPredictor, dynamics, dynamics_masked
    from adaptigraph_amd import construct_edges_from_states_batch, pad_torch, truncate_graph, decode_action

Hand-written HIP kernels (adaptigraph_amd/csrc) behind a C-ABI (include/adaptigraph_amd.h).  No CPU fallback.
"""
from .context import Engine, default_engine
from .forward_dynamics import dynamics, dynamics_masked, rollout_work
from .graph import (EdgeList, construct_edges_from_states_batch, construct_edges_from_states, construct_edges_index,
                    construct_edges_with_backoff, pad_torch, truncate_graph)
from .model import DynamicsPredictor
from .plan_utils import decode_action
from .losses import chamfer, mean_chamfer, box_loss, rope_penalty, cloth_penalty, granular_penalty
from .costs import running_cost
from .physics_param_optimizer import dynamics_error, dynamics_error_sweep
from .mppi import angle_normalize, clip_actions, sample_action_seq, optimize_action_mppi, mpc_iteration
from .planner import Planner

__all__ = ["Engine", "default_engine", "dynamics", "dynamics_masked", "rollout_work", "EdgeList", "construct_edges_from_states_batch", "construct_edges_from_states",
           "construct_edges_index", "construct_edges_with_backoff", "pad_torch", "truncate_graph", "DynamicsPredictor", "decode_action", "chamfer",
           "mean_chamfer", "box_loss", "rope_penalty", "cloth_penalty", "granular_penalty", "running_cost", "dynamics_error", "dynamics_error_sweep", "angle_normalize",
           "clip_actions", "sample_action_seq", "optimize_action_mppi", "mpc_iteration", "Planner"]
