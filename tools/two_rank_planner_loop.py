"""The reference's UNCHANGED planner loop (src/planning/plan.py:210, 241-247: `planner.total_chunks = n_chunk`, n_chunk x
`trajectory_optimization`, `merge_res`) on the SHIPPED configuration (20,000 pushes as 40 chunks of 500, n_look_ahead 1) with
planner_config['group'] set: call ci runs on rank ci % world (on that rank's side streams), the other ranks draw its samples
only; merge_res all-gathers the 40 winners and broadcasts the best one's outputs (adaptigraph_amd/planner.py: _rank_dealt,
_merge_rank_dealt).  Prints one JSON line (rank 0): SHA-256 of the merged result (act_seq, best rollout, its reward) - which must
not depend on the number of ranks -, ms per planner call (max over ranks), the calls each rank rolled out and whether every rank's
generator ended in the same state.

  python tools/two_rank_planner_loop.py                                   (one rank)
  AG_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/two_rank_planner_loop.py                                     (two ranks on one GPU, gloo)
env: AG_LOOP_MATERIAL (rope), AG_LOOP_CHUNKS (40), AG_LOOP_SAMPLES (500), AG_LOOP_UPDATE_ITER (1), AG_LOOP_REPS (3)
Diagnostic / test driver; the contract line is bench.py's."""
import hashlib, json, os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AG_BENCH_SHARE_GPU") == "1":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AG_BENCH_BACKEND", "gloo" if os.environ.get("AG_BENCH_SHARE_GPU") == "1" else "nccl"))
    import bench_configs as BC
    import bench_planner as BP
    dev = torch.device("cuda", local)
    BP.dev = BC.dev = dev
    mat = os.environ.get("AG_LOOP_MATERIAL", "rope")
    n_chunk = int(os.environ.get("AG_LOOP_CHUNKS", "40"))
    S = int(os.environ.get("AG_LOOP_SAMPLES", "500"))
    reps = int(os.environ.get("AG_LOOP_REPS", "3"))
    planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, S, np.random.default_rng(0))
    planner.n_update_iter = int(os.environ.get("AG_LOOP_UPDATE_ITER", "1"))
    if world > 1:
        planner.group = True                                                 # = planner_config['group']
    eng = m.engine(dev)
    torch.manual_seed(0)
    act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    for _ in range(2):
        torch.manual_seed(1)
        BP.loop_call(planner, s0, act_seq, n_chunk)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.manual_seed(1)
        res = BP.loop_call(planner, s0, act_seq, n_chunk)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gen = hashlib.sha256(torch.cuda.get_rng_state(dev).numpy().tobytes()).hexdigest()
    h = hashlib.sha256()
    for t in (res["act_seq"], res["best_model_output"]["state_seqs"], res["best_model_output"]["action_seqs"], res["best_eval_output"]["reward_seqs"]):
        h.update(t.detach().cpu().numpy().tobytes())
    mine = {"rank": rank, "ms_per_planner_call": ms, "generator_sha256": gen, "result_sha256": h.hexdigest(),
            "calls_owned": len([k for k in range(n_chunk) if k % world == rank])}
    per = [mine]
    if world > 1:
        per = [None] * world
        dist.all_gather_object(per, mine)
    if rank == 0:
        print(json.dumps({"material": mat, "world": world, "n_chunk": n_chunk, "n_sample": S, "n_update_iter": planner.n_update_iter,
                          "result_sha256": per[0]["result_sha256"], "same_result_on_every_rank": len({p["result_sha256"] for p in per}) == 1,
                          "generators_in_step": len({p["generator_sha256"] for p in per}) == 1, "generator_sha256": per[0]["generator_sha256"],
                          "ms_per_planner_call": max(p["ms_per_planner_call"] for p in per), "calls_owned_per_rank": [p["calls_owned"] for p in per],
                          "best_reward": float(res["best_eval_output"]["reward_seqs"].mean()),
                          "backend": dist.get_backend() if world > 1 else None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
