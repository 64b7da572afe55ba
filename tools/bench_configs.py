"""Throughput of the BASELINE configs that are parity cases, not the bench line (BASELINE.json configs[0..2, 4]):
rope 1 x 10, rope 64 x 20, granular 256 x 20, mixed 512 x 20 (variable-size graphs through dynamics_masked-style
padding, one model context per material).  Synthetic clouds as SURVEY 8(d); random-init weights.  One GPU.
Prints one JSON object per config.  Diagnostic: the contract line is bench.py's."""
import json, os, sys, time, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                            # random_weights, model_cfg shapes
import adaptigraph_amd as ag

dev = torch.device("cuda", 0)
TASKS = {
    "rope": dict(adj_thresh=0.5, topk=10, connect_tools_all=False, push_length=0.1, gripper_enable=False, eef_num=1,
                 pusher_points=[[0.0, 0.0, 0.12]]),
    "granular": dict(adj_thresh=0.4, topk=20, connect_tools_all=False, push_length=0.2, gripper_enable=False, eef_num=5,
                     pusher_points=[[0, 0, 0.1], [0, 0.05, 0.1], [0, 0.025, 0.1], [0, -0.025, 0.1], [0, -0.05, 0.1]]),
    "cloth": dict(adj_thresh=0.75, topk=5, connect_tools_all=True, push_length=0.1, gripper_enable=True, eef_num=1,
                  pusher_points=[[0.0, 0.0, 0.17]]),
}


def task_of(mat, N_o):
    t = dict(sim_real_ratio=10, max_n=1, n_his=4, material=mat, material_dims={mat: 1}, material_indices={mat: 0})
    t.update(TASKS[mat])
    t["max_nR"] = int(1.2 * (t["topk"] + t["eef_num"]) * (N_o + t["eef_num"])) + 64
    return t


def cloud_of(mat, rng):
    if mat == "rope":
        t = np.linspace(0, 1, 300)
        p = np.stack([-2 + 3 * t, 0 * t, 0.5 * np.sin(6 * t)], 1)
        return (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)
    side, pitch, jit = (32, 0.12, 0.02) if mat == "granular" else (45, 0.3, 0.02)
    g = (np.arange(side) - (side - 1) / 2.0) * pitch
    xx, zz = np.meshgrid(g, g, indexing="ij")
    p = np.stack([xx.ravel() - 2.0, np.zeros(side * side), zz.ravel() + 1.0], 1)
    return (p + rng.normal(0, jit, p.shape)).astype(np.float32)


def model_of(mat):
    mc, _, _ = B.model_cfg()
    m = ag.DynamicsPredictor(mc, {"material_index": {mat: 0}, mat: {"physics_params": [{"name": "p", "use": True}]}},
                             {"n_his": 4, "materials": [mat]}, dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in B.random_weights(0).items()})
    return m


def ppm_of(task, mat):
    return types.SimpleNamespace(task_config=task, eef_num=task["eef_num"], material=mat, material_dims=task["material_dims"],
                                 material_indices=task["material_indices"], physics_param={mat: torch.tensor([0.5])},
                                 adj_thresh=task["adj_thresh"])


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def homogeneous(mat, Bn, H, R, reps):
    rng = np.random.default_rng(0)
    cloud = cloud_of(mat, rng)
    task = task_of(mat, cloud.shape[0])
    task["action_upper_lim"] = [0.0, 4.5, 3.14, float(R)]          # bounds the repeat: GPU-resident actions take ag_rollout_actions
    m, s0 = model_of(mat), torch.from_numpy(cloud).to(dev)
    a = torch.from_numpy(B.make_actions(Bn, H, R, cloud, rng)).to(dev)
    ppm = ppm_of(task, mat)
    dt = timed(lambda: ag.dynamics(s0, a, m, dev, ppm), reps)
    return {"config": f"{mat} {cloud.shape[0]}+{task['eef_num']} particles, {Bn} candidates x {H * R} steps",
            "ms_per_call": dt * 1e3, "rollout_steps_per_s": Bn * H * R / dt}


def mixed(total, steps, reps):
    """cfg 5: a third of the batch per material, every candidate with its own particle count U{N/2..N} (padded + masked);
    dynamics_masked advances one look-ahead step of `steps` repeats."""
    rng = np.random.default_rng(1)
    calls, n_steps = [], 0
    for mat, nb in (("rope", total // 3 + total % 3), ("granular", total // 3), ("cloth", total // 3)):
        cloud = cloud_of(mat, rng)
        N = cloud.shape[0]
        task = task_of(mat, N)
        m, ppm = model_of(mat), ppm_of(task_of(mat, N), mat)
        state = np.repeat(cloud[None], nb, 0)
        mask = np.zeros((nb, N), bool)
        for b in range(nb):
            mask[b, :rng.integers(N // 2, N + 1)] = True
        a = B.make_actions(nb, 1, steps, cloud, rng)[:, 0]
        args = (torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a).to(dev))
        calls.append(lambda args=args, m=m, ppm=ppm: ag.dynamics_masked(*args, m, dev, ppm))
        n_steps += nb * steps
    dt = timed(lambda: [c() for c in calls], reps)
    return {"config": f"mixed rope+granular+cloth, {total} variable-size graphs x {steps} steps", "ms_per_call": dt * 1e3,
            "rollout_steps_per_s": n_steps / dt}


if __name__ == "__main__":
    for r in (homogeneous("rope", 1, 1, 10, 20), homogeneous("rope", 64, 2, 10, 10), homogeneous("granular", 256, 2, 10, 3),
              mixed(512, 20, 3)):
        print(json.dumps(r))
