"""Work-balanced shards of the SHIPPED planner workload over the ranks of a torch.distributed group (reference
src/planning/plan.py:177-247 draws 20,000 pushes uniformly over the action box, plan_utils.py:48-50, and evaluates them on
ONE device, plan.py:87).  Every rank draws the same batch (same seed), calls adaptigraph_amd.rollout_work on the FULL batch
(forwards left per candidate once the contact-free prefix is taken from one base rollout), cuts contiguous shards by work
(sharding.work_balanced_bounds), rolls out and evaluates its shard, and the rewards are all-gathered.  Prints one JSON line
(rank 0): reward SHA-256 of the work-balanced run, of the count-balanced run and (1 rank) of the unsharded evaluation - all
three must agree bit for bit - plus per-rank candidates, forwards executed and wall time for both cuts.

  python tools/two_rank_planner_shards.py                                  (one rank: the unsharded hash)
  AG_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/two_rank_planner_shards.py                                    (two ranks on one GPU, gloo)
Diagnostic / test driver; the contract line is bench.py's."""
import hashlib, json, os, sys, time
from functools import partial
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
    import adaptigraph_amd as ag
    import bench_planner as BP
    from adaptigraph_amd.sharding import sharded_candidate_rewards, work_balanced_bounds, shard_bounds
    mat = os.environ.get("AG_SHARD_MATERIAL", "rope")
    B = int(os.environ.get("AG_SHARD_CANDIDATES", "4000"))
    dev = torch.device("cuda", local)
    BP.dev = dev
    planner, m, s0, lo, hi, cloud, task = BP.make_planner(mat, B, np.random.default_rng(0))
    ppm = planner.model_rollout.keywords["ppm_optimizer"]
    eng = m.engine(dev)
    torch.manual_seed(7)                                                    # every rank draws the same batch
    act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    actions = planner.sample_action_sequences(act_seq, iter_index=0)
    group = True if world > 1 else None
    tgt_fn = planner.evaluate_traj                                           # running_cost bound to the task (rank-local maxima)
    ev = partial(ag.running_cost, error_func=tgt_fn.keywords["error_func"], penalty_func=tgt_fn.keywords["penalty_func"],
                 bbox=tgt_fn.keywords["bbox"], group=group)
    stats = {}

    def run(label, work_fn):
        seen = {}

        def rollout(a):
            t0 = time.perf_counter()
            out = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
            torch.cuda.synchronize()
            seen["ms"] = (time.perf_counter() - t0) * 1e3
            seen["n"] = int(a.shape[0])
            seen["executed"] = eng.rollout_counts()[0]
            return out

        def reward(seq, a):
            return ev(seq, a, state_cur=s0)["reward_seqs"]

        for _ in range(2):                                                  # (second pass: warm workspaces, kept base rollout)
            r = sharded_candidate_rewards(actions, rollout, reward, group=None if world == 1 else None, work_fn=work_fn)
        per = [seen]
        if world > 1:
            per = [None] * world
            dist.all_gather_object(per, seen)
        stats[label] = per
        return hashlib.sha256(r.detach().cpu().numpy().tobytes()).hexdigest()

    work_fn = lambda a: ag.rollout_work(s0, a, m, dev, ppm)
    sha_w = run("work_balanced", work_fn)
    sha_c = run("count_balanced", None)
    work = work_fn(actions)
    if rank == 0:
        wb = work_balanced_bounds(work, world)
        print(json.dumps({"material": mat, "candidates": B, "world": world, "reward_sha256_work_balanced": sha_w,
                          "reward_sha256_count_balanced": sha_c, "never_touch_fraction": float((work == 0).mean()),
                          "work_total_forwards": int(work.sum()),
                          "work_balanced": {"bounds": wb, "forwards_per_rank": [int(work[a:b].sum()) for a, b in wb], "per_rank": stats["work_balanced"]},
                          "count_balanced": {"bounds": [shard_bounds(B, world, r) for r in range(world)],
                                             "forwards_per_rank": [int(work[a:b].sum()) for a, b in (shard_bounds(B, world, r) for r in range(world))],
                                             "per_rank": stats["count_balanced"]}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
