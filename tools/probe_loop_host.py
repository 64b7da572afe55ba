"""Where the shipped planner call's time goes on the HOST, unprofiled (rocprofv3 makes a launch cost ~28 us and serialises the
loop): per planner call of 40 x trajectory_optimization (tools/bench_planner.py's workload) the time until the last call has
been ENQUEUED (host) against the time until the GPU has finished (total), for the class defaults and variants.  One JSON line
per (material, variant).  Diagnostic; one GPU."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_planner as BP

dev = torch.device("cuda", 0)


def main():
    mats = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rope"]
    for mat in mats:
        planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, 500, np.random.default_rng(0))
        eng = m.engine(dev)
        torch.manual_seed(0)
        act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
        for label, pipe, reuse, opts in (("default", 6, True, {}), ("4 side streams", 4, True, {}), ("8 side streams", 8, True, {}), ("streams=1", 6, True, {"streams": 1}),
                                         ("pipeline_fork=1", 6, True, {"pipeline_fork": 1}),
                                         ("one stream, no wait (pipeline 1)", 1, True, {}), ("r04", 0, False, {})):
            planner.pipeline_chunks, planner.reuse_best_rollout = pipe, reuse
            with eng.options(**opts):
                for _ in range(2):
                    torch.manual_seed(1); BP.loop_call(planner, s0, act_seq, 40)
                torch.cuda.synchronize()
                host, total = [], []
                for _ in range(5):
                    torch.manual_seed(1)
                    t0 = time.perf_counter()
                    res_all = []
                    planner.total_chunks = 40
                    for ci in range(40):
                        planner.chunk_id = ci
                        res_all.append(planner.trajectory_optimization(s0, act_seq))
                    t1 = time.perf_counter()
                    planner.merge_res(res_all)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
                print(json.dumps({"material": mat, "variant": label, "host_enqueue_ms": float(np.median(host)),
                                  "planner_call_ms": float(np.median(total))}), flush=True)


if __name__ == "__main__":
    main()
