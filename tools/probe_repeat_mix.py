"""Does the shrinking live prefix of a repeat-sorted chunk cost throughput?  One dynamics() call over 20,000 rope candidates
(200 + 1 particles, the shipped planner's size) with (a) the shipped mix of repeats 5..14 and (b) one repeat count for all
(same total of candidate-forwards to within 1 %): microseconds per candidate-forward.  Diagnostic, one GPU."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench as B
import adaptigraph_amd as ag
from bench_configs import TASKS, model_of, ppm_of
from bench_planner import cloud_of, LIMITS

dev = torch.device("cuda", 0)
for mat in ("rope", "granular"):
    rng = np.random.default_rng(0)
    cloud = cloud_of(mat, rng)
    t = dict(sim_real_ratio=10, max_n=1, n_his=4, material=mat, material_dims={mat: 1}, material_indices={mat: 0})
    t.update(TASKS[mat])
    t["max_nR"] = int(1.2 * (t["topk"] + t["eef_num"]) * (cloud.shape[0] + t["eef_num"])) + 64
    t["action_lower_lim"], t["action_upper_lim"] = LIMITS[mat]
    m, ppm = model_of(mat), ppm_of(t, mat)
    s0 = torch.from_numpy(cloud).to(dev)
    lo, hi = LIMITS[mat][0][3], LIMITS[mat][1][3]
    n = 20000
    for label, lens in (("mixed", rng.uniform(lo, hi, n)), ("sorted", np.sort(rng.uniform(lo, hi, n))[::-1].copy()),
                        ("constant", np.full(n, (lo + hi) / 2 + 0.25))):
        a = B.make_actions(n, 1, 1, cloud, rng)
        a[:, 0, 3] = lens
        a = torch.from_numpy(a.astype(np.float32)).to(dev)
        eng = m.engine(dev)
        for _ in range(2):
            ag.dynamics(s0, a, m, dev, ppm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ag.dynamics(s0, a, m, dev, ppm)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ex, need = eng.rollout_counts()
        enq, bound = eng.launch_counts()
        print(json.dumps({"material": mat, "repeats": label, "ms_per_call": dt * 1e3, "candidate_forwards": need,
                          "us_per_candidate_forward": dt * 1e6 / need, "steps_enqueued": enq, "steps_bound": bound}), flush=True)
