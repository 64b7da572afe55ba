"""Per-kernel-family HIP-event times of one small rollout (diagnostic): where a launch-bound batch spends its time."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC
import bench as B
import adaptigraph_amd as ag
from adaptigraph_amd import _lib

dev = torch.device("cuda", 0)
for mat, Bn, H, R in (("rope", 1, 1, 10), ("rope", 64, 2, 10), ("cloth", 1, 2, 10)):
    rng = np.random.default_rng(0)
    cloud = BC.cloud_of(mat, rng)
    task = BC.task_of(mat, cloud.shape[0])
    m, s0 = BC.model_of(mat), torch.from_numpy(cloud).to(dev)
    a = torch.from_numpy(B.make_actions(Bn, H, R, cloud, rng)).to(dev)
    ppm = BC.ppm_of(task, mat)
    eng = m.engine(dev)
    fams = [f for f in _lib.KERNEL_FAMILIES if f not in ("mp", "prep", "cost")]
    ag.dynamics(s0, a, m, dev, ppm); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ag.dynamics(s0, a, m, dev, ppm)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5
    eng.reset_stats(); eng.set_profiling(fams)
    ag.dynamics(s0, a, m, dev, ppm); torch.cuda.synchronize()
    eng.set_profiling([])
    st = {f: eng.kernel_stats(f) for f in fams}
    print(json.dumps({"config": f"{mat} {cloud.shape[0]} particles, {Bn} x {H*R} steps", "wall_ms": wall * 1e3,
                      "kernel_us_per_launch": {f: round(v[0] / max(1, v[1]) * 1e3, 1) for f, v in st.items()},
                      "launches": {f: v[1] for f, v in st.items()},
                      "sum_kernel_ms": sum(v[0] for v in st.values())}))
