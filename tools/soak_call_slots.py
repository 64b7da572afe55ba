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
    # r06: random_interact.py's configuration (five update rounds per call; lone calls waiting once, dealt calls whose later rounds
    # skip the prefix census) against the strict execution, 1 / 2 / 3 chunks
    import bench_interact as BI
    n_interact = 0
    for mat in ("rope", "granular", "cloth"):
        for n_chunk in (1, 2, 3):
            planner, m, s0, lo, hi, cloud, task = BI.make_interact_planner(mat, 1000 // n_chunk, np.random.default_rng(0))
            for seed in range(max(1, n_seeds // 3)):
                torch.manual_seed(200 + seed)
                act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
                outs = {}
                for label, pipe, reuse in (("default", 6, True), ("strict", 0, False)):
                    planner.pipeline_chunks, planner.reuse_best_rollout = pipe, reuse
                    torch.manual_seed(seed)
                    res = BP.loop_call(planner, s0, act_seq, n_chunk)
                    torch.cuda.synchronize()
                    outs[label] = (res["act_seq"].clone(), res["best_model_output"]["state_seqs"].clone(),
                                   res["best_eval_output"]["reward_seqs"].clone(), torch.cuda.get_rng_state(dev))
                if not all(torch.equal(a, b) for a, b in zip(outs["default"], outs["strict"])):
                    sys.exit(f"MISMATCH interact {mat} chunks {n_chunk} seed {seed}")
                n_interact += 1
    # r06: dynamics_mixed against the sequential dynamics_masked calls, random sub-batches of three materials
    import bench_configs as BC
    n_mixed = 0
    parts = []
    for mat in ("rope", "granular", "cloth"):
        c = BC.cloud_of(mat, np.random.default_rng(3))
        parts.append((mat, c, BC.model_of(mat), BC.ppm_of(BC.task_of(mat, c.shape[0]), mat)))
    for rep in range(n_seeds):
        batches = []
        for mat, c, mm, pp in parts:
            nb = int(rng.integers(1, 40))
            N = c.shape[0]
            mask = np.zeros((nb, N), bool)
            for b in range(nb):
                mask[b, :rng.integers(N // 2, N + 1)] = True
            import bench as B
            a = B.make_actions(nb, 1, int(rng.integers(1, 6)), c, rng)[:, 0]
            batches.append((torch.from_numpy(np.repeat(c[None], nb, 0)).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a), mm, pp))
        want = [ag.dynamics_masked(b[0], b[1], b[2], b[3], dev, b[4])["state_seqs"].clone() for b in batches]
        got = ag.dynamics_mixed(batches, dev)
        if not all(torch.equal(g_["state_seqs"], w) for g_, w in zip(got, want)):
            sys.exit(f"MISMATCH mixed rep {rep}")
        n_mixed += 1
    print(json.dumps({"ok": True, "comparisons": checked + n_interact + n_mixed, "planner_loop_and_async_calls": checked,
                      "interact_configuration_calls": n_interact, "mixed_batches": n_mixed, "seconds": time.time() - t0}))


if __name__ == "__main__":
    main()
