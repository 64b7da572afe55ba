"""Where does one planner call (trajectory_optimization_chunked, shipped configuration) spend its time?  Wall time per phase with a
device synchronisation after each (so the phases add up to a little more than the pipelined call).  Diagnostic, one GPU."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_planner as BP

dev = torch.device("cuda", 0)
for mat in sys.argv[1:] or ["rope", "granular", "cloth"]:
    rng = np.random.default_rng(0)
    planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, 500, rng)
    torch.manual_seed(0)
    act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    T = {}
    def timed(name, fn):
        def wrap(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrap
    for _ in range(2):
        planner.trajectory_optimization_chunked(s0, act_seq, 40)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    planner.trajectory_optimization_chunked(s0, act_seq, 40)
    torch.cuda.synchronize(); whole = time.perf_counter() - t0
    planner.sample_action_sequences = timed("sample (40 calls)", planner.sample_action_sequences)
    planner.model_rollout = timed("rollout (20000 candidates, then the 40 winners)", planner.model_rollout)
    planner.evaluate_traj = timed("evaluate (40 x 500 candidates, then 40 x 1)", planner.evaluate_traj)
    planner.merge_res = timed("merge_res", planner.merge_res)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    planner.trajectory_optimization_chunked(s0, act_seq, 40)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    print(json.dumps({"material": mat, "ms_pipelined": whole * 1e3, "ms_with_syncs": total * 1e3,
                      "phases_ms": {k: round(v * 1e3, 2) for k, v in T.items()}}), flush=True)
