"""Does capturing a small rollout in a hipGraph shorten it?  One rope graph x 10 steps (BASELINE configs[0]) and the planner's
batch-of-one re-roll of a cloth-2026 candidate x 20 steps, device-planned actions (the call then enqueues a fixed sequence of
kernels on the caller's stream and touches the host nowhere): eager call latency vs replay of a torch.cuda.CUDAGraph that
captured the same call.  Prints one JSON object.  Diagnostic."""
import json, os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import adaptigraph_amd as ag
import bench_configs as BC

dev = torch.device("cuda", 0)
out = {}
for mat, R, H in (("rope", 10, 1), ("cloth", 10, 2)):
    rng = np.random.default_rng(0)
    cloud = BC.cloud_of(mat, rng)
    task = BC.task_of(mat, cloud.shape[0])
    task["action_upper_lim"] = [0.0, 4.5, 3.14, float(R)]
    m, ppm = BC.model_of(mat), BC.ppm_of(task, mat)
    s0 = torch.from_numpy(cloud).to(dev)
    a = torch.from_numpy(BC.B.make_actions(1, H, R, cloud, rng)).to(dev)
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    call = lambda: ag.dynamics(s0, a, m, dev, ppm, _sync=False, _overflow_flag=flag)["state_seqs"]
    for _ in range(5):
        ref = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):                            # warm-up on the capture stream (workspace, lazily created state)
        for _ in range(3):
            call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            got = call()
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(got, ref))
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 50 * 1e3
        out[f"{mat}_1x{H * R}"] = {"eager_ms_per_call": eager, "graph_replay_ms_per_call": replay, "replay_equals_eager": same}
    except Exception as e:  # noqa: BLE001
        out[f"{mat}_1x{H * R}"] = {"eager_ms_per_call": eager, "capture_failed": repr(e)[:300]}
print(json.dumps(out))
