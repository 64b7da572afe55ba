"""dynamics_error: the objective the online physics-parameter optimiser evaluates for every BO / CMA-ES proposal
(reference src/planning/physics_param_optimizer.py:178-226) = masked rollout of the past interactions + mean chamfer
to what really happened.  Same signature; dynamics_masked and mean_chamfer run on the HIP engine.  The host optimisers
(skopt.gp_minimize / cma) stay in the reference.  r05: dynamics_error_sweep evaluates a population / sweep of
parameters in one go (independent evaluations dealt to streams, one read-back)."""
from __future__ import annotations

import copy

import numpy as np
import torch

from .context import side_streams as _sweep_streams
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


def _problem(ppm_optimizer, state_init_list, state_real_list, actions):
    """the padded clouds, masks and pushes of :187-215, uploaded once"""
    device = ppm_optimizer.device
    rows = ppm_optimizer.task_config["max_nobj"]
    before, before_valid = _pad_clouds(state_init_list[:len(actions)], rows, device)
    after, after_valid = _pad_clouds(state_real_list[:len(actions)], rows, device)
    pushes = torch.from_numpy(np.stack(actions, axis=0).astype(np.float32))     # CPU: the decode runs on the host (bit-equal cos / sin)
    return before, before_valid, after, after_valid, pushes


def _as_param_dict(physics_param, ppm_optimizer):
    if isinstance(physics_param, (list, np.ndarray)):
        names = list(ppm_optimizer.material_dims.keys())
        assert len(names) == 1, "only support single material now"
        return {names[0]: torch.tensor(np.asarray(physics_param), dtype=torch.float32)}
    return copy.deepcopy(physics_param)


def dynamics_error(physics_param, ppm_optimizer, state_init_list, state_real_list, actions):
    """physics_param: a list / array of values for the single material (what gp_minimize / cma hand over, :183-186) or a
    {material: tensor} dict; state_*_list: per past interaction the observed cloud before / after the push; actions: the pushes.
    -> mean over the interactions of the masked chamfer distance between the predicted and the observed cloud (:219-226)."""
    device = ppm_optimizer.device
    physics_param = _as_param_dict(physics_param, ppm_optimizer)
    before, before_valid, after, after_valid, pushes = _problem(ppm_optimizer, state_init_list, state_real_list, actions)
    rolled = dynamics_masked(before, before_valid, pushes, ppm_optimizer.model, device, ppm_optimizer, physics_param=physics_param)
    return mean_chamfer(rolled["state_seqs"].detach(), after, before_valid, after_valid).mean()


@torch.no_grad()
def dynamics_error_sweep(physics_params, ppm_optimizer, state_init_list, state_real_list, actions, streams=4):
    """dynamics_error for a LIST of physics parameters - a CMA-ES population (`es.ask()` of optimize_cma's strategy, :125-175:
    its members are independent) or a sweep - in one go: (K,) float64 numpy, element k equal to
    dynamics_error(physics_params[k], ...) bit for bit.  The problem (padded clouds, masks, pushes) is uploaded once; evaluation k
    is enqueued on side stream k % streams without waiting for anything (dynamics_masked(_sync=False): an evaluation is <= 20
    small graphs, i.e. a chain of latency-bound launches that leaves the chip mostly idle - several of them run side by side on
    the context's per-stream call slots); one read-back at the end brings all chamfer values and overflow flags.  Raises
    Exception("Exceeds max dims") if any evaluation's graph exceeded max_nR, as the reference's dynamics_masked would have."""
    from .losses import chamfer
    device = torch.device(ppm_optimizer.device)
    before, before_valid, after, after_valid, pushes = _problem(ppm_optimizer, state_init_list, state_real_list, actions)
    K, n = len(physics_params), before.shape[0]
    if K == 0:
        return np.zeros(0, np.float64)
    errs = torch.empty((K, n), device=device, dtype=torch.float32)
    flags = torch.zeros((K, 2), device=device, dtype=torch.int32)
    cur = torch.cuda.current_stream(device)
    entry = torch.cuda.Event()
    entry.record(cur)
    side = _sweep_streams(device, max(1, min(int(streams), K)))
    for k, pp in enumerate(physics_params):
        st = side[k % len(side)]
        if k < len(side):
            st.wait_event(entry)
        with torch.cuda.stream(st):
            rolled = dynamics_masked(before, before_valid, pushes, ppm_optimizer.model, device, ppm_optimizer,
                                     physics_param=_as_param_dict(pp, ppm_optimizer), _sync=False, _overflow_flag=flags[k])
            errs[k] = chamfer(rolled["state_seqs"], after, before_valid, after_valid)
            rolled["state_seqs"].record_stream(st)
    for st in side:
        done = torch.cuda.Event()
        done.record(st)
        cur.wait_event(done)
    host = torch.cat([errs, flags.to(torch.float32)], 1).cpu().numpy()           # the one wait
    if (host[:, n] > float(ppm_optimizer.task_config["max_nR"])).any():
        raise Exception("Exceeds max dims")                                      # utils.py:63-65
    return host[:, :n].astype(np.float64).mean(1)                                # :225 on mean_chamfer's float64 array
