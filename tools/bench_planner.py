"""The SHIPPED planner workload (reference src/planning/plan.py:177-247 with src/config/planning/{rope,granular,cloth}.yaml):
n_sample 20000 evaluated as 40 chunks of n_sample_chunk 500, n_look_ahead 1, n_update_iter 1, push length drawn from the
yaml's action limits (rope: U[5,15) -> action_repeat 5..14; granular / cloth: U[2,10) -> 2..9), max_nobj 200 object
particles.  Times one planner call (= one outer MPC iteration's "get action" block, plan.py:241-247):
  loop     the reference's 40-iteration host loop of Planner.trajectory_optimization + merge_res (plus variants, see main)
  chunked  Planner.trajectory_optimization_chunked (one rollout call for all 20000 candidates, one for the 40 winners)
each with the repeat-aware launch order on (default) and off (option repeat_sort 0: every candidate of a launch chunk is
stepped to the chunk's maximum repeat, as the reference steps the batch).  Prints one JSON object per (material, mode).
Diagnostic: the contract line is bench.py's.  One GPU."""
import json, os, sys, time, types
from functools import partial
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench as B
import adaptigraph_amd as ag
from adaptigraph_amd.planner import Planner
from bench_configs import TASKS, model_of, ppm_of

dev = torch.device("cuda", 0)
LIMITS = {"rope": ([-4.5, -2.5, -3.14, 5.0], [0.0, 4.5, 3.14, 15.0]),          # planning/rope.yaml:28-29
          "granular": ([-4.5, -2.5, -3.14, 2.0], [0.0, 4.5, 3.14, 10.0]),      # planning/granular.yaml:32-33
          "cloth": ([-4.5, -2.5, -3.14, 2.0], [0.0, 4.5, 3.14, 10.0])}         # planning/cloth.yaml:28-29
BBOX = np.array([[-0.45, 0.0], [-0.25, 0.45]]) * 10.0                           # yaml bbox x sim_real_ratio (plan.py:170-174)


def cloud_of(mat, rng):
    """max_nobj = 200 object particles (planning/*.yaml), inside the action box"""
    if mat == "rope":
        t = np.linspace(0, 1, 200)
        p = np.stack([-3.5 + 2.5 * t, 0 * t, 1.0 + 0.5 * np.sin(6 * t)], 1)
        return (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)
    pitch = 0.12 if mat == "granular" else 0.3
    g = (np.arange(14) - 6.5) * pitch
    xx, zz = np.meshgrid(g, g, indexing="ij")
    p = np.stack([xx.ravel() - 2.2, np.zeros(196), zz.ravel() + 1.0], 1)
    return (p + rng.normal(0, 0.02, p.shape)).astype(np.float32)


def make_planner(mat, n_sample, rng):
    cloud = cloud_of(mat, rng)
    t = dict(sim_real_ratio=10, max_n=1, n_his=4, material=mat, material_dims={mat: 1}, material_indices={mat: 0})
    t.update(TASKS[mat])
    t["max_nR"] = int(1.2 * (t["topk"] + t["eef_num"]) * (cloud.shape[0] + t["eef_num"])) + 64   # shipped 2000 < E here
    lo, hi = (torch.tensor(v, device=dev) for v in LIMITS[mat])
    t["action_lower_lim"], t["action_upper_lim"] = LIMITS[mat]
    m, ppm = model_of(mat), ppm_of(t, mat)
    s0 = torch.from_numpy(cloud).to(dev)
    target = torch.from_numpy(cloud + np.float32([0.4, 0, 0.3])).to(dev)
    pen = {"rope": ag.rope_penalty, "granular": ag.granular_penalty, "cloth": ag.cloth_penalty}[mat]
    cfg = {"action_dim": 4,
           "model_rollout_fn": partial(ag.dynamics, model=m, device=dev, ppm_optimizer=ppm),
           "evaluate_traj_fn": partial(ag.running_cost, error_func=partial(ag.chamfer, y=target[None]),
                                       penalty_func=partial(pen, sim_real_ratio=10.0), bbox=BBOX),
           "sampling_action_seq_fn": partial(ag.sample_action_seq, action_lower_lim=lo, action_upper_lim=hi, n_sample=n_sample,
                                             device=dev, noise_level=1.0, push_length=t["push_length"]),
           "clip_action_seq_fn": partial(ag.clip_actions, action_lower_lim=lo, action_upper_lim=hi),
           "optimize_action_mppi_fn": partial(ag.optimize_action_mppi, reward_weight=500.0, action_lower_lim=lo,
                                              action_upper_lim=hi, push_length=t["push_length"]),
           "n_sample": n_sample, "n_look_ahead": 1, "n_update_iter": 1, "reward_weight": 500.0, "action_lower_lim": lo,
           "action_upper_lim": hi, "planner_type": "MPPI", "device": dev, "verbose": False, "noise_level": 1.0,
           "rollout_best": True}
    return Planner(cfg), m, s0, lo, hi, cloud, t


def loop_call(planner, s0, act_seq, n_chunk):
    res_all = []
    planner.total_chunks = n_chunk                                       # plan.py:210
    for ci in range(n_chunk):                                            # plan.py:241-247
        planner.chunk_id = ci
        res = planner.trajectory_optimization(s0, act_seq)
        res_all.append({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in res.items()})
    return planner.merge_res(res_all)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--materials", default="rope,granular,cloth")
    ap.add_argument("--modes", default="loop,chunked,loop_r04,loop_nopipe,loop_reroll,chunked_reroll")
    ap.add_argument("--sorts", default="1,0")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n_sample, n_chunk, reps = 500, 40, args.reps
    for mat in args.materials.split(","):
        rng = np.random.default_rng(0)
        planner, m, s0, lo, hi, cloud, task = make_planner(mat, n_sample, rng)
        eng = m.engine(dev)
        torch.manual_seed(0)
        act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
        def variant(fn, pipeline, reuse):
            def run():
                old = planner.pipeline_chunks, planner.reuse_best_rollout
                planner.pipeline_chunks, planner.reuse_best_rollout = pipeline, reuse
                try:
                    return fn()
                finally:
                    planner.pipeline_chunks, planner.reuse_best_rollout = old
            return run
        loop_fn = lambda: loop_call(planner, s0, act_seq, n_chunk)
        chunked_fn = lambda: planner.trajectory_optimization_chunked(s0, act_seq, n_chunk)
        # loop / chunked: the class as a drop-in gets it (r05: independent calls dealt to 6 streams, winners' rollouts taken out of
        # their batches).  loop_r04: one stream, every call waits for its flags, winners re-rolled with a batch of one (the r04
        # behaviour).  loop_nopipe / loop_reroll: one of the two r05 changes each.  chunked_reroll: winners re-rolled (one call).
        # (the strict one-stream modes run FIRST: HIP maps streams to hardware queues when they are created, and in a process that
        # has already created the six side streams the in-library stream a strict call forks onto sometimes lands on a queue that
        # is shared - the same strict loop then takes 366 instead of 263 ms, run-to-run random; option streams=1 removes it)
        for mode, fn in (("loop_r04", variant(loop_fn, 0, False)), ("loop_nopipe", variant(loop_fn, 0, True)),
                         ("loop", variant(loop_fn, 6, True)), ("chunked", variant(chunked_fn, 6, True)),
                         ("loop_reroll", variant(loop_fn, 6, False)), ("loop_2streams", variant(loop_fn, 2, True)),
                         ("loop_4streams", variant(loop_fn, 4, True)), ("chunked_reroll", variant(chunked_fn, 6, False))):
            if mode not in args.modes.split(","):
                continue
            for sort in [int(x) for x in args.sorts.split(",")]:
                if mode not in ("loop", "chunked") and sort == 0:
                    continue
                with eng.options(repeat_sort=sort):
                    torch.manual_seed(1)
                    fn()
                    torch.cuda.synchronize()
                    ex = need = 0
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        torch.manual_seed(1)
                        res = fn()
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                    # candidate-forwards of the big rollout call(s): re-run one sampling + rollout of all candidates
                    torch.manual_seed(1)
                    a = torch.cat([planner.sample_action_sequences(act_seq, iter_index=0) for _ in range(n_chunk)])
                    if mode.startswith("loop"):
                        for ci in range(n_chunk):
                            planner.model_rollout(s0, a[ci * n_sample:(ci + 1) * n_sample])
                            e, n = eng.rollout_counts()
                            ex, need = ex + e, need + n
                    else:
                        planner.model_rollout(s0, a)
                        ex, need = eng.rollout_counts()
                print(json.dumps({"config": f"{mat} {cloud.shape[0]}+{task['eef_num']} particles, planner call: {n_chunk} chunks x "
                                            f"{n_sample} candidates, n_look_ahead 1, action_repeat {int(LIMITS[mat][0][3])}.."
                                            f"{int(np.ceil(LIMITS[mat][1][3])) - 1}",
                                  "mode": mode, "repeat_sort": sort, "ms_per_planner_call": dt * 1e3,
                                  "candidate_forwards_executed": int(ex), "candidate_forwards_needed": int(need),
                                  # effective: the candidate-forwards the reference's loop would run (every candidate stepped
                                  # action_repeat times) over this time; executed: what the engine actually ran (prefix sharing)
                                  "effective_rollout_steps_per_s": need / dt, "executed_rollout_steps_per_s": ex / dt,
                                  "best_reward": float(res["best_eval_output"]["reward_seqs"].mean())}), flush=True)


if __name__ == "__main__":
    main()
