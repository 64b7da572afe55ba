"""Bounded soak of the per-stream call slots and the planner's dealt loop (one process, one GPU): for several seeds the shipped
40 x 500 planner call is run dealt to six streams and strictly on one stream and the results compared bit for bit (winner, its
rollout, its reward, generator state), on all three materials; then asynchronous dynamics() calls of random shapes over eleven
streams against the synchronous results.  Prints one JSON line; exits non-zero on the first mismatch.  Diagnostic."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_planner as BP
import adaptigraph_amd as ag

dev = torch.device("cuda", 0)


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    t0 = time.time()
    checked = 0
    for mat in ("rope", "granular", "cloth"):
        planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, 500, np.random.default_rng(0))
        for seed in range(n_seeds):
            torch.manual_seed(100 + seed)
            act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
            outs = {}
            for label, pipe, reuse in (("dealt", 6, True), ("strict", 0, False), ("dealt3", 3, False)):
                planner.pipeline_chunks, planner.reuse_best_rollout = pipe, reuse
                torch.manual_seed(seed)
                res = BP.loop_call(planner, s0, act_seq, 40)
                torch.cuda.synchronize()
                outs[label] = (res["act_seq"].clone(), res["best_model_output"]["state_seqs"].clone(),
                               res["best_eval_output"]["reward_seqs"].clone(), torch.cuda.get_rng_state(dev))
            for label in ("dealt", "dealt3"):
                if not all(torch.equal(a, b) for a, b in zip(outs[label], outs["strict"])):
                    sys.exit(f"MISMATCH {mat} seed {seed} {label}")
            checked += 1
    # random shapes over eleven streams
    rng = np.random.default_rng(5)
    planner, m, s0, lo, hi, cloud, task = BP.make_planner("rope", 500, np.random.default_rng(0))
    ppm = planner.model_rollout.keywords["ppm_optimizer"]
    streams = [torch.cuda.Stream(dev) for _ in range(11)]
    for rep in range(n_seeds):
        jobs = []
        for i in range(22):
            B = int(rng.integers(1, 700))
            a = planner.sample_action_sequences(torch.rand((1, 4), device=dev) * (hi - lo) + lo, iter_index=0)[:B].clone()
            jobs.append(a)
        want = [ag.dynamics(s0, a, m, dev, ppm)["state_seqs"].clone() for a in jobs]
        torch.cuda.synchronize()
        flags = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in jobs]
        got = []
        for i, a in enumerate(jobs):
            with torch.cuda.stream(streams[(i * 7 + rep) % 11]):
                got.append(ag.dynamics(s0, a, m, dev, ppm, _sync=False, _overflow_flag=flags[i])["state_seqs"])
        torch.cuda.synchronize()
        for i in range(len(jobs)):
            if not torch.equal(got[i], want[i]):
                sys.exit(f"MISMATCH async job {i} rep {rep} B {jobs[i].shape[0]}")
        checked += len(jobs)
    print(json.dumps({"ok": True, "comparisons": checked, "seconds": time.time() - t0}))


if __name__ == "__main__":
    main()
