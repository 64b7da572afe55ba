"""The planner configuration of the reference's SECOND Planner call site, src/planning/random_interact.py:155-218: n_look_ahead 1,
n_update_iter 5, n_sample 1000, noise_level 1.0, reward_weight 1000, evaluated as n_sample / n_sample_chunk chunks - the file
ships n_sample_chunk 1000 (ONE trajectory_optimization per planner call); 500 (two chunks) is timed beside it.  Per material and
chunking, one planner call (`planner.total_chunks = n_chunk`, the loop, merge_res) in two executions:
  strict   config['pipeline_chunks'] 0, winners re-rolled with a batch of one: every rollout waits for its flags (the r04 behaviour)
  default  what the class does by itself: a lone call enqueues its five rounds and waits ONCE before it returns; two chunks are
           dealt to side streams and waited for in merge_res; the best-so-far selection between rounds stays on the device
and whether both give the same bits.  Prints one JSON object per (material, chunking).  Diagnostic: the contract line is bench.py's."""
import hashlib, json, os, sys, time
from functools import partial
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_planner as BP
import adaptigraph_amd as ag


def make_interact_planner(mat, n_sample, rng):
    planner, m, s0, lo, hi, cloud, t = BP.make_planner(mat, n_sample, rng)
    cfg = dict(planner.config)
    dev = BP.dev
    cfg.update({"n_update_iter": 5, "reward_weight": 1000.0, "noise_level": 1.0,                      # random_interact.py:164-183
                "sampling_action_seq_fn": partial(ag.sample_action_seq, action_lower_lim=lo, action_upper_lim=hi, n_sample=n_sample,
                                                  device=dev, noise_level=1.0, push_length=t["push_length"]),
                "optimize_action_mppi_fn": partial(ag.optimize_action_mppi, reward_weight=1000.0, action_lower_lim=lo,
                                                   action_upper_lim=hi, push_length=t["push_length"])})
    from adaptigraph_amd.planner import Planner
    return Planner(cfg), m, s0, lo, hi, cloud, t


def _sha(res):
    h = hashlib.sha256()
    for x in (res["act_seq"], res["best_model_output"]["state_seqs"], res["best_eval_output"]["reward_seqs"]):
        h.update(x.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--materials", default="rope,granular,cloth")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = BP.dev
    for mat in args.materials.split(","):
        for n_chunk in (1, 2):
            S = 1000 // n_chunk
            planner, m, s0, lo, hi, cloud, task = make_interact_planner(mat, S, np.random.default_rng(0))
            torch.manual_seed(0)
            act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
            out = {}
            for label, pipe, reuse, later in (("strict", 0, False, False), ("default", 6, True, False), ("prefix_later", 6, True, True)):
                planner.pipeline_chunks, planner.reuse_best_rollout, planner._dealt_prefix_later = pipe, reuse, later
                for _ in range(2):
                    torch.manual_seed(1); BP.loop_call(planner, s0, act_seq, n_chunk)
                torch.cuda.synchronize()
                ts = []
                for _ in range(args.reps):
                    torch.manual_seed(1)
                    t0 = time.perf_counter()
                    res = BP.loop_call(planner, s0, act_seq, n_chunk)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                out[label] = (float(np.median(ts)), _sha(res), float(res["best_eval_output"]["reward_seqs"].mean()))
            print(json.dumps({"config": f"{mat} {cloud.shape[0]}+{task['eef_num']} particles, random_interact.py planner call: n_update_iter 5, "
                                        f"n_sample 1000 as {n_chunk} chunk(s) of {S}, n_look_ahead 1",
                              "ms_per_planner_call_strict": out["strict"][0], "ms_per_planner_call_default": out["default"][0],
                              "ms_per_planner_call_dealt_with_prefix_in_later_rounds": out["prefix_later"][0],
                              "bit_equal": out["strict"][1] == out["default"][1] == out["prefix_later"][1], "best_reward": out["default"][2],
                              "default_is": "one call, one wait at its end" if n_chunk == 1 else "two calls dealt to side streams, waited for in merge_res"}),
                  flush=True)


if __name__ == "__main__":
    main()
