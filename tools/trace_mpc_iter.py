"""One MPC iteration with GPU-resident actions (sample -> rollout -> running_cost -> MPPI update -> best candidate re-rolled),
for a HIP-API trace:  rocprofv3 --hip-trace --kernel-trace --output-format csv -d DIR -o t -- python3 tools/trace_mpc_iter.py
then  python tools/trace_mpc_iter.py --report DIR  lists every HIP API call between the sampling kernel (k_mppi_sample) and
the first kernel of the rollout that consumes its output, for the device-planned path (task config bounds the repeat:
ag_rollout_actions) and for the host-decode path (option device_decode 0).  Without rocprofv3 it prints call latencies of
dynamics() for one rope graph x 10 steps and the planner's chunk (500 candidates) on both paths."""
import csv, glob, json, os, sys, time
import numpy as np


def report(d):
    def rows(pat):
        out = []
        for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
            out += list(csv.DictReader(open(f)))
        return out
    api, ker = rows("*hip_api_trace.csv"), rows("*kernel_trace.csv")
    ker.sort(key=lambda r: int(r["Start_Timestamp"]))
    api.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_corr = {r["Correlation_Id"]: r for r in api}
    samples = [k for k in ker if "k_mppi_sample" in k["Kernel_Name"]]
    res = []
    for s in samples:
        nxt = next((k for k in ker if int(k["Start_Timestamp"]) > int(s["Start_Timestamp"]) and
                    ("k_roll_plan" in k["Kernel_Name"] or "k_roll_init" in k["Kernel_Name"])), None)
        if nxt is None:
            continue
        a0, a1 = by_corr.get(s["Correlation_Id"]), by_corr.get(nxt["Correlation_Id"])
        if not a0 or not a1:
            continue
        between = [r["Function"] for r in api if int(a0["End_Timestamp"]) <= int(r["Start_Timestamp"]) <= int(a1["Start_Timestamp"])]
        blocking = [f for f in between if f in ("hipMemcpy", "hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize")
                    or (f.startswith("hipMemcpy") and "Async" not in f)]
        res.append({"first_rollout_kernel": nxt["Kernel_Name"].split("(")[0], "host_us_between_launches":
                    (int(a1["Start_Timestamp"]) - int(a0["End_Timestamp"])) / 1e3, "gpu_idle_us_between_kernels":
                    (int(nxt["Start_Timestamp"]) - int(s["End_Timestamp"])) / 1e3, "hip_calls_between": between,
                    "blocking_calls_between": blocking})
    # the last two (warm) iterations of each path
    out = [r for kind in ("k_roll_plan", "k_roll_init") for r in [x for x in res if kind in x["first_rollout_kernel"]][-2:]]
    print(json.dumps(out, indent=1))


def main():
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import adaptigraph_amd as ag
    import bench_planner as BP
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    planner, m, s0, lo, hi, cloud, task = BP.make_planner("rope", 500, rng)
    eng = m.engine(dev)
    roll, ev = planner.model_rollout, planner.evaluate_traj
    torch.manual_seed(0)
    act0 = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    out = {}
    for mode in (-1, 0):                                     # device-planned, then host decode
        with eng.options(device_decode=mode):
            for _ in range(3):
                ag.mpc_iteration(s0, act0, roll, ev, lo, hi, 500, dev, push_length=task["push_length"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ag.mpc_iteration(s0, act0, roll, ev, lo, hi, 500, dev, push_length=task["push_length"])
            torch.cuda.synchronize()
            out[f"ms_per_mpc_iteration_500_candidates_device_decode_{mode}"] = (time.perf_counter() - t0) / 5 * 1e3
            one = torch.tensor([[[-2.0, 1.2, 0.4, 10.5]]], device=dev)           # one graph x 10 steps (BASELINE configs[0])
            for _ in range(5):
                roll(s0, one)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                roll(s0, one)
            torch.cuda.synchronize()
            out[f"ms_per_call_rope_1x10_device_decode_{mode}"] = (time.perf_counter() - t0) / 50 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        main()
