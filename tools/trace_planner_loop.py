"""GPU-busy vs idle per iteration of the shipped planner call (reference src/planning/plan.py:241-247: 40 x
Planner.trajectory_optimization + merge_res), from a rocprofv3 HIP-API + kernel trace.

  cd /tmp && rocprofv3 --hip-trace --kernel-trace --output-format csv -d DIR -o t -- python3 $R/tools/trace_planner_loop.py run rope loop
  python tools/trace_planner_loop.py report DIR > profiles/r05_planner_loop_trace_rope.json

`run` executes 2 warm planner calls and 1 traced one (mode loop | loop_r04 | loop_nopipe | chunked); `report` cuts the LAST planner call of
the trace into its iterations at the sampling kernels (k_mppi_sample: one per trajectory_optimization call) and prints, per
call and per iteration (median / total): wall time, the union of kernel intervals (GPU busy), the idle rest, kernels launched,
and the time the host spent inside blocking HIP calls.  Diagnostic tool; one GPU.
Caveat: under rocprofv3 a kernel launch costs the host ~28 us (5-6 us unprofiled), so a loop of small launches becomes launch-bound
and the side streams of the default (dealt) mode hardly ever hold work at the same time - its trace shows ~1 kernel in flight
(--kernel-trace alone: 250 ms per call, 143 unprofiled).  The trace is the right tool for the one-stream modes (is the GPU busy or
idle behind the waits?) and for the chunked entry; the dealt mode is measured unprofiled by tools/probe_loop_host.py."""
import csv, glob, json, os, sys
import numpy as np


def _rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


BLOCKING = ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyDtoH", "hipMemcpyHtoD",
            "hipMemcpyWithStream")


def report(d):
    ker = _rows(d, "*kernel_trace.csv")
    api = _rows(d, "*hip_api_trace.csv")
    ker = [(int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"].split("(")[0].replace("void ", ""),
            (k.get("Queue_Id"), k.get("Stream_Id"))) for k in ker]
    ker.sort()
    api = [(int(a["Start_Timestamp"]), int(a["End_Timestamp"]), a["Function"]) for a in api]
    api.sort()
    samples = [k for k in ker if "k_mppi_sample" in k[2]]
    meta = {}
    mp = os.path.join(d, "run_meta.json")
    if os.path.exists(mp):
        meta = json.load(open(mp))
    n_iter = int(os.environ.get("AG_TRACE_ITERS", meta.get("sample_kernels_per_call", 40)))
    mode = meta.get("mode", "loop")
    if mode.startswith("chunked"):
        # all 40 chunks are sampled back to back, then ONE rollout call: the call starts at the 40th-last sampling kernel
        t_begin = samples[-n_iter][0]
        cuts = [t_begin, ker[-1][1] + 1]
    else:
        t_begin = samples[-n_iter][0]
        cuts = [s[0] for s in samples[-n_iter:]] + [ker[-1][1] + 1]
    per = []
    blk_n = {}
    for i in range(len(cuts) - 1):
        a, b = cuts[i], cuts[i + 1]
        ks = [k for k in ker if a <= k[0] < b]
        busy, cur_s, cur_e = 0, None, None
        for s, e, _, _ in ks:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        blk = {}
        for s, e, f in api:
            if a <= s < b and f in BLOCKING:
                blk[f] = blk.get(f, 0) + (e - s)
                blk_n[f] = blk_n.get(f, 0) + 1
        by_k = {}
        for s, e, n, _ in ks:
            by_k[n] = by_k.get(n, 0) + (e - s)
        gaps = sorted(((ks[j + 1][0] - max(k[1] for k in ks[:j + 1][-8:]), ks[j][2], ks[j + 1][2]) for j in range(len(ks) - 1)), reverse=True)[:3]
        per.append({"wall_us": (b - a) / 1e3, "gpu_busy_us": busy / 1e3, "gpu_idle_us": (b - a - busy) / 1e3, "kernels": len(ks),
                    "host_blocked_us": {k: v / 1e3 for k, v in blk.items()}, "kernel_us": {k: v / 1e3 for k, v in by_k.items()},
                    "largest_gaps_us": [(g[0] / 1e3, g[1], g[2]) for g in gaps]})
    tot = lambda key: float(sum(p[key] for p in per))
    kern = {}
    for p in per:
        for k, v in p["kernel_us"].items():
            kern[k] = kern.get(k, 0.0) + v
    blk = {}
    for p in per:
        for k, v in p["host_blocked_us"].items():
            blk[k] = blk.get(k, 0.0) + v
    top = dict(sorted(kern.items(), key=lambda kv: -kv[1])[:12])
    # concurrency over the whole call: kernel time summed over all kernels / union of their intervals; per (queue, stream) shares
    call_k = [k for k in ker if cuts[0] <= k[0] < cuts[-1]]
    ksum = sum(e - s for s, e, _, _ in call_k)
    by_q = {}
    for s, e, _, q in call_k:
        by_q[str(q)] = by_q.get(str(q), 0) + (e - s)
    # host side: time inside HIP API calls (any) and the longest of them
    api_call = [a for a in api if cuts[0] <= a[0] < cuts[-1]]
    api_by = {}
    for s0_, e0_, f in api_call:
        v = api_by.setdefault(f, [0, 0.0]); v[0] += 1; v[1] += (e0_ - s0_) / 1e6
    api_top = dict(sorted(api_by.items(), key=lambda kv: -kv[1][1])[:10])
    out = {"what": "one planner call (last of the trace) cut at the sampling kernels; times in us (per-iteration) / ms (totals)",
           "run": meta, "iterations": len(per),
           "call_ms": tot("wall_us") / 1e3, "gpu_busy_ms": tot("gpu_busy_us") / 1e3, "gpu_idle_ms": tot("gpu_idle_us") / 1e3,
           "gpu_busy_fraction": tot("gpu_busy_us") / max(1e-9, tot("wall_us")),
           "kernels_launched": int(tot("kernels")), "kernel_time_sum_ms": ksum / 1e6,
           "mean_kernels_in_flight_while_busy": ksum / 1e6 / max(1e-9, tot("gpu_busy_us") / 1e3),
           "kernel_ms_by_queue_and_stream": {k: v / 1e6 for k, v in by_q.items()},
           "host_hip_api_calls_top_ms": {k: {"calls": v[0], "ms": v[1]} for k, v in api_top.items()},
           "host_blocked_ms": {k: v / 1e3 for k, v in blk.items()},
           # blocking HIP calls the host made between the first sampling kernel of the call and its last kernel (merge_res's one
           # read-back included): hipStreamSynchronize / hipEventSynchronize / hipMemcpy* / hipDeviceSynchronize
           "host_blocking_calls": blk_n, "host_blocking_calls_total": int(sum(blk_n.values())),
           "per_iteration_median_us": {k: float(np.median([p[k] for p in per])) for k in ("wall_us", "gpu_busy_us", "gpu_idle_us", "kernels")},
           "kernel_time_ms_top": {k: v / 1e3 for k, v in top.items()},
           "example_iteration": per[len(per) // 2] if per else None}
    print(json.dumps(out, indent=1))


def run(mat, mode):
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_planner as BP
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    n_chunk, per_call = 40, 40
    if mode.startswith("interact"):
        # random_interact.py:155-214: n_update_iter 5, n_sample 1000; interact1 = its n_sample_chunk 1000 (ONE call per planner call),
        # interact2 = two chunks of 500; suffix _strict = every rollout waited for, winners re-rolled (the r04 behaviour)
        import bench_interact as BI
        n_chunk = 1 if mode.startswith("interact1") else 2
        planner, m, s0, lo, hi, cloud, task = BI.make_interact_planner(mat, 1000 // n_chunk, rng)
        per_call = n_chunk * planner.n_update_iter
        if mode.endswith("_strict"):
            planner.pipeline_chunks, planner.reuse_best_rollout = 0, False
    else:
        planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, 500, rng)
    torch.manual_seed(0)
    act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    if mode == "loop_r04":                                  # one stream, every call waits, winners re-rolled
        planner.pipeline_chunks, planner.reuse_best_rollout = 0, False
    elif mode == "loop_nopipe":
        planner.pipeline_chunks = 0
    elif mode not in ("loop", "chunked") and not mode.startswith("interact"):
        raise SystemExit("mode: loop | loop_r04 | loop_nopipe | chunked | interact1[_strict] | interact2[_strict]")
    fn = (lambda: BP.loop_call(planner, s0, act_seq, n_chunk)) if not mode.startswith("chunked") else \
         (lambda: planner.trajectory_optimization_chunked(s0, act_seq, 40))
    import time
    for _ in range(2):
        torch.manual_seed(1)
        fn()
    torch.cuda.synchronize()
    torch.manual_seed(1)
    t0 = time.perf_counter()
    res = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    meta = {"material": mat, "mode": mode, "traced_call_ms_host_clock": dt * 1e3, "sample_kernels_per_call": per_call,
            "best_reward": float(res["best_eval_output"]["reward_seqs"].mean())}
    out_dir = os.environ.get("AG_TRACE_META_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        json.dump(meta, open(os.path.join(out_dir, "run_meta.json"), "w"))
    print(json.dumps(meta))


if __name__ == "__main__":
    if sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run(sys.argv[2], sys.argv[3])
