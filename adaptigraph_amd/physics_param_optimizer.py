"""dynamics_error: the objective the online physics-parameter optimiser evaluates for every BO / CMA-ES proposal
(reference src/planning/physics_param_optimizer.py:178-226) = masked rollout of the past interactions + mean chamfer
to what really happened.  Same signature; dynamics_masked and mean_chamfer run on the HIP engine.  The host optimisers
(skopt.gp_minimize / cma) stay in the reference."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .forward_dynamics import dynamics_masked
from .losses import mean_chamfer


def _pad_clouds(clouds, rows, device):
    """list of (n_i, 3) arrays -> ((len, rows, 3) float32, (len, rows) bool) on `device`: cloud i in the first n_i rows"""
    pad = np.zeros((len(clouds), rows, 3), np.float32)
    valid = np.zeros((len(clouds), rows), bool)
    for i, cloud in enumerate(clouds):
        pad[i, :len(cloud)] = cloud
        valid[i, :len(cloud)] = True
    return torch.from_numpy(pad).to(device), torch.from_numpy(valid).to(device)


def dynamics_error(physics_param, ppm_optimizer, state_init_list, state_real_list, actions):
    """physics_param: a list / array of values for the single material (what gp_minimize / cma hand over, :183-186) or a
    {material: tensor} dict; state_*_list: per past interaction the observed cloud before / after the push; actions: the pushes.
    -> mean over the interactions of the masked chamfer distance between the predicted and the observed cloud (:219-226)."""
    device = ppm_optimizer.device
    if isinstance(physics_param, (list, np.ndarray)):
        names = list(ppm_optimizer.material_dims.keys())
        assert len(names) == 1, "only support single material now"
        physics_param = {names[0]: torch.tensor(np.asarray(physics_param), dtype=torch.float32)}
    else:
        physics_param = copy.deepcopy(physics_param)
    rows = ppm_optimizer.task_config["max_nobj"]
    before, before_valid = _pad_clouds(state_init_list[:len(actions)], rows, device)
    after, after_valid = _pad_clouds(state_real_list[:len(actions)], rows, device)
    pushes = torch.from_numpy(np.stack(actions, axis=0).astype(np.float32)).to(device)
    rolled = dynamics_masked(before, before_valid, pushes, ppm_optimizer.model, device, ppm_optimizer, physics_param=physics_param)
    return mean_chamfer(rolled["state_seqs"].detach(), after, before_valid, after_valid).mean()
