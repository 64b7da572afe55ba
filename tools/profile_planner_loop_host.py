import cProfile, pstats, io, sys, os, time
import numpy as np, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_planner as BP
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
planner, m, s0, lo, hi, cloud, task = BP.make_planner("rope", 500, rng)
torch.manual_seed(0)
act = torch.rand((1, 4), device=dev) * (hi - lo) + lo
for _ in range(2):
    BP.loop_call(planner, s0, act, 40)
torch.cuda.synchronize()
t0 = time.perf_counter(); BP.loop_call(planner, s0, act, 40); torch.cuda.synchronize(); print("loop call ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable()
BP.loop_call(planner, s0, act, 40)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
