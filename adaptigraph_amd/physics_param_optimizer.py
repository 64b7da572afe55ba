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


def dynamics_error(physics_param, ppm_optimizer, state_init_list, state_real_list, actions):
    len_act = len(actions)
    physics_param = copy.deepcopy(physics_param)
    device = ppm_optimizer.device
    if isinstance(physics_param, (list, np.ndarray)):                                  # :183-186
        assert len(list(ppm_optimizer.material_dims.keys())) == 1, "only support single material now"
        material_name = list(ppm_optimizer.material_dims.keys())[0]
        physics_param = {material_name: torch.tensor(physics_param, dtype=torch.float32)}
    max_nobj = ppm_optimizer.task_config["max_nobj"]
    init_mask = np.zeros((len_act, max_nobj), bool)
    final_mask = np.zeros((len_act, max_nobj), bool)
    init_pad = np.zeros((len_act, max_nobj, 3), np.float32)
    final_pad = np.zeros((len_act, max_nobj, 3), np.float32)
    for i in range(len_act):                                                           # :196-208
        ni, nf = state_init_list[i].shape[0], state_real_list[i].shape[0]
        init_mask[i, :ni] = True
        final_mask[i, :nf] = True
        init_pad[i, :ni] = state_init_list[i]
        final_pad[i, :nf] = state_real_list[i]
    state_init_all = torch.from_numpy(init_pad).to(device)
    state_init_mask = torch.from_numpy(init_mask).to(device)
    state_final_all = torch.from_numpy(final_pad).to(device)
    state_final_mask = torch.from_numpy(final_mask).to(device)
    acts = torch.from_numpy(np.stack(actions, axis=0).astype(np.float32)).to(device)
    out = dynamics_masked(state_init_all, state_init_mask, acts, ppm_optimizer.model, device, ppm_optimizer,
                          physics_param=physics_param)                                 # :219-220
    error = mean_chamfer(out["state_seqs"].detach(), state_final_all, state_init_mask, state_final_mask)   # :223
    return error.mean()
