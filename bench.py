#!/usr/bin/env python3
"""Headline benchmark: rollout-steps/s of the GNN-dynamics rollout (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3]): cloth, 1024 MPC candidates x 20 rollout steps
(2 look-ahead steps x action_repeat 10) x 2025 object particles + 1 gripper particle, graph rebuilt every
step, random-init weights, synthetic jittered 45x45 cloth.  One "step" of this bench = one dynamics() call over
the whole candidate batch + the planner's running_cost (chamfer to a target cloud, cloth penalty, bbox penalty)
(+ RCCL all-reduce(MAX) of two scalars and all-gather of the per-candidate rewards when N > 1).
Candidates are independent, so they are sharded over ranks (STRONG scaling: the 1024-candidate batch is fixed);
the only collective is the all-gather of B/N fp32 costs per rank.
"""
import argparse
import json
import os
import sys
import time
import types


def _parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--candidates", type=int, default=1024)
    ap.add_argument("--side", type=int, default=45, help="cloth grid side (45 -> 2025 particles)")
    ap.add_argument("--lookahead", type=int, default=2)
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="HIP-event time every kernel family (adds overhead)")
    ap.add_argument("--no-kernel-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-bf16x3", action="store_true", help="skip the secondary bf16x3-mode measurement")
    ap.add_argument("--host-decode", action="store_true", help="decode the actions on the host (ag_rollout) instead of the "
                    "device-planned path the planner's GPU-resident samples take (ag_rollout_actions)")
    ap.add_argument("--no-mpc-iter", action="store_true", help="skip the ms/MPC-iteration leg (profiling runs: its B=1 "
                    "best-candidate rollouts would dilute per-kernel averages)")
    ap.add_argument("--plumbing-only", default=None, metavar="ok|fail-rank-R",
                    help="launch check without a GPU (tests/test_bench_launch.py): the ranks meet over gloo, rank 0 prints what "
                         "every rank saw of the launch, nothing of the engine is imported; fail-rank-R makes rank R exit 3")
    return ap


def launch_command(n_gpus, argv, port):
    """The command `bench.py --gpus N` starts when no launcher wrapped it: the driver's documented launch line."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks_if_needed(argv):
    """`python bench.py --gpus N` with N > 1 and NO launcher in front (no RANK / WORLD_SIZE in the environment): this process - which
    has imported nothing that touches the GPU - starts the N ranks as a CHILD `python -m torch.distributed.run ... bench.py <same
    args>` (never an exec), lets the child write to this process's stdout / stderr (rank 0's JSON line arrives verbatim) and exits
    with the child's code, non-zero if any rank failed.  Under an existing launcher (WORLD_SIZE set) it returns and the caller
    runs as one rank.  Returns None to carry on in this process, else the exit code."""
    args, _ = _parser().parse_known_args(argv)
    if args.gpus <= 1 or ("WORLD_SIZE" in os.environ and "RANK" in os.environ):
        return None
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["AG_BENCH_LAUNCHED_BY"] = "bench.py"
    cmd = launch_command(args.gpus, argv, port)
    print("bench.py: --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, cwd=os.path.dirname(os.path.abspath(__file__))).returncode


if __name__ == "__main__":
    _rc = launch_ranks_if_needed(sys.argv[1:])          # before torch / HIP: the parent of the ranks never initialises the GPU
    if _rc is not None:
        sys.exit(_rc)

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The engine runs a large batch on four
# streams; torch.distributed adds RCCL's, and with five streams on four queues two of the engine's share one and serialise
# (measured with a one-rank RCCL group: 474.8 ms per step against 466.2 without the group; with 8 queues 467.5 / 465.7).  Read by
# the HIP runtime when it initialises, i.e. after this line.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FLOPs of the formulation the kernels execute (factored W_rp / W_pp; SURVEY §8(d) F_min), padding not counted
FLOP_PER_EDGE = 2 * (17 * 150 + 2 * 150 * 150) + 2 * 150 * 150                     # relation encoder + W1  = 140,100
FLOP_PER_NODE_ENC = 2 * (6 * 150 + 2 * 150 * 150) + 3 * 2 * 150 * 150              # particle encoder + Wa,W2,W3
FLOP_PER_NODE_PROP = 3 * 2 * 150 * 150                                             # Wb + W2 + W3 (per round)
FLOP_PER_NODE_FINAL = 2 * 150 * 150 + 2 * (2 * 150 * 150 + 3 * 150)                # Wb + predictor
PEAK_FP32_MFMA_TFLOPS = 157.3                                                      # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0                                                              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3                                                 # MI355X_MICROARCH.md: ~2.5 PF dense, 16x the fp32 MFMA rate
# outputs of the IMPORTED REFERENCE for candidates of this bench's own batch (tests/golden/make_golden.py --fullsize[-r05]):
# data only - state0, actions, weights (or their SHA-256), state_seqs; nothing of the reference runs here.  full_cloth_seqs (r05)
# covers EVERY candidate parity_check looks at (parity_picks(1024, 64)), with the reference's own smallest edge-selection margin
# per look-ahead step beside its outputs; the two older files hold four of those candidates with per-forward records.
REFERENCE_FIXTURES = ("full_cloth_seqs", "full_cloth_a", "full_cloth_flip")


def random_weights(seed, nf=150, in_dim=6, rel_dim=17):
    """nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) init of the 22 state_dict tensors (no checkpoint exists
    offline: SURVEY §4).  Local on purpose: the timed path must not touch oracle/."""
    rng = np.random.default_rng(seed)
    shapes = {"particle_encoder.model.0": (nf, in_dim), "particle_encoder.model.2": (nf, nf),
              "particle_encoder.model.4": (nf, nf), "relation_encoder.model.0": (nf, rel_dim),
              "relation_encoder.model.2": (nf, nf), "relation_encoder.model.4": (nf, nf),
              "particle_propagator.linear": (nf, 2 * nf), "relation_propagator.linear": (nf, 3 * nf),
              "non_rigid_predictor.linear_0": (nf, nf), "non_rigid_predictor.linear_1": (nf, nf),
              "non_rigid_predictor.linear_2": (3, nf)}
    W = {}
    for k, (o, i) in shapes.items():
        bound = 1.0 / np.sqrt(i)
        W[k + ".weight"] = rng.uniform(-bound, bound, (o, i)).astype(np.float32)
        W[k + ".bias"] = rng.uniform(-bound, bound, (o,)).astype(np.float32)
    return W


def cloth_cloud(side, rng):
    g = (np.arange(side) - (side - 1) / 2.0) * 0.3
    xx, zz = np.meshgrid(g, g, indexing="ij")
    p = np.stack([xx.ravel() - 2.0, np.zeros(side * side), zz.ravel() + 1.0], 1)
    return (p + rng.normal(0, 0.02, p.shape)).astype(np.float32)


def make_task(max_nR, limits=True):
    t = dict(adj_thresh=0.75, topk=5, connect_tools_all=True, sim_real_ratio=10, push_length=0.1,
             gripper_enable=True, max_n=1, max_nR=max_nR, n_his=4, eef_num=1, material="cloth",
             pusher_points=[[0.0, 0.0, 0.170]], material_dims={"cloth": 1}, material_indices={"cloth": 0})
    if limits:   # planning/cloth.yaml:28-29.  With them in the task config GPU-resident actions take the device-planned path
        t.update(action_lower_lim=[-4.5, -2.5, -3.14, 2.0], action_upper_lim=[0.0, 4.5, 3.14, 10.0])   # (ag_rollout_actions)
    return t


def model_cfg():
    mc = dict(verbose=False, nf_particle=150, nf_relation=150, nf_effect=150, nf_physics=10, attr_dim=2, state_dim=0,
              offset_dim=0, action_dim=3, density_dim=0, pstep=3, sequence_len=4, rel_particle_dim=0, rel_attr_dim=2,
              rel_group_dim=1, rel_distance_dim=3, rel_density_dim=0)
    mat = {"material_index": {"cloth": 0}, "cloth": {"physics_params": [{"name": "sf", "use": True}]}}
    return mc, mat, {"n_his": 4, "materials": ["cloth"]}


def make_actions(B, H, repeat, cloud, rng):
    a = np.zeros((B, H, 4), np.float32)
    c = cloud.mean(0)
    a[..., 0] = c[0] + rng.uniform(-2.0, 2.0, (B, H))
    a[..., 1] = c[2] + rng.uniform(-2.0, 2.0, (B, H))
    a[..., 2] = rng.uniform(-3.14, 3.14, (B, H))
    a[..., 3] = repeat + 0.5
    return a


def parity_picks(n_local, n_pick):
    """candidates of the timed batch the CPU leg looks at: first, last and evenly spaced ones (every launch chunk of every
    stream is hit).  tests/golden/make_golden.py --fullsize-r05 runs the REFERENCE on exactly this list for the default batch."""
    return sorted({int(round(i * (n_local - 1) / max(1, n_pick - 1))) for i in range(n_pick)})


def _sha256(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def reference_golden(cloud, task, W, actions, with_margin=False):
    """{candidate id: state_seqs (H,N_o,3)} recorded from the reference itself for candidates of THIS batch: a fixture counts
    only if its start state, weights, task scalars and the candidate's raw action are bit-equal to what this run times (the
    compact r05 file stores SHA-256 digests of start state and weights instead of the arrays).  with_margin: values are
    (state_seqs, margin) with margin (H,) = the reference's own smallest edge-selection margin over the forwards of each
    look-ahead step, or None where the fixture does not hold it.  The first fixture that covers a candidate serves it."""
    out = {}
    for name in REFERENCE_FIXTURES:
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        gt = json.loads(bytes(g["task_json"]).decode())
        same_task = all(gt.get(k) == task.get(k) for k in ("adj_thresh", "topk", "connect_tools_all", "sim_real_ratio", "push_length",
                                                           "gripper_enable", "n_his", "eef_num", "pusher_points"))
        if not same_task or int(g["pstep"]) != 3:
            continue
        if "sha_state0" in g.files:
            if not np.array_equal(g["sha_state0"], _sha256(cloud)):
                continue
            if any(("sha_w::" + k) not in g.files or not np.array_equal(g["sha_w::" + k], _sha256(v)) for k, v in W.items()):
                continue
        else:
            if g["state0"].shape != cloud.shape or not np.array_equal(g["state0"], cloud):
                continue
            if any(("w::" + k) not in g.files or not np.array_equal(g["w::" + k], v) for k, v in W.items()):
                continue
        margin = g["reference_margin"] if "reference_margin" in g.files else None
        for j, cid in enumerate(g["cand_ids"]):
            if int(cid) in out:
                continue
            if 0 <= cid < len(actions) and g["action"][j].shape == actions[cid].shape and np.array_equal(g["action"][j], actions[cid]):
                seq = np.asarray(g["state_seqs"][j])
                out[int(cid)] = (seq, None if margin is None else np.asarray(margin[j])) if with_margin else seq
    return out


def _oracle_child(cloud, task, W, actions, workers, blas_threads):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src, dst = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(src, cloud=cloud, actions=actions, pstep=3,
                 task_json=np.frombuffer(json.dumps(task).encode(), dtype=np.uint8),
                 **{"w::" + k: v for k, v in W.items()})
        subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", src, dst, str(workers), str(blas_threads)], check=True, cwd=ROOT)
        z = np.load(dst)
        return z["state_seqs"], float(z["seconds"]), int(z["steps"]), z["margin"]


def cpu_baseline(cloud, task, W, actions, picks, gpu_seqs, ref, tol=1e-5):
    """CPU leg (after the timed region; the only place the oracle is used): the numpy oracle (CPU restatement of the
    reference, oracle/) rolls out `picks` - candidates OF THE TIMED BATCH - one per worker process and BLAS thread, in
    a child process (this one holds the GPU and must not fork).  Returns (cpu_baseline, parity_check): the timing of
    that bounded sample and the comparison of what the GPU produced for the same candidates in the last timed step with
    (1) the REFERENCE's own outputs where a committed fixture holds them (`ref`: candidate -> state_seqs) and (2) the oracle."""
    workers = max(1, min(len(picks), 16, os.cpu_count() or 1))      # a one-GPU box has a 16-core CPU share
    want, dt, steps, margin = _oracle_child(cloud, task, W, actions[picks], workers, 1)
    # Per candidate and look-ahead step.  A free-running rollout can only be compared while both sides build the same
    # graph: once the oracle itself passes an edge decision that a position change within the tolerance would flip
    # (selection margin < 4*adj_thresh*tol), a deviation from then on is a different-but-valid graph, not an error
    # (tests/test_gpu_fullsize.py checks such flips pair by pair; here they are reported, not hidden).
    err = np.abs(gpu_seqs - want).reshape(len(picks), want.shape[1], -1).max(-1)          # (P, H)
    tie_margin = 4.0 * float(task["adj_thresh"]) * tol
    tie_prone = np.minimum.accumulate(margin, axis=1) < tie_margin
    within = err <= tol
    post_flip_bound = 1e-3                                   # a flipped edge moves a particle by ~1e-4..1e-3 over the rest of the rollout
    unexplained = ~within & (~tie_prone | (err > post_flip_bound))
    clean = within.all(1)
    # ---- the reference itself, where a fixture holds its outputs for a candidate of this batch (r05: all of `picks`)
    ref_ids = [c for c in picks if c in ref]
    vs_ref = None
    ref_within = {}
    ref_unexplained = False
    if ref_ids:
        rerr = {c: np.abs(gpu_seqs[picks.index(c)] - ref[c][0]).reshape(want.shape[1], -1).max(-1) for c in ref_ids}    # (H,) each
        per = {c: float(e.max()) for c, e in rerr.items()}
        ref_within = {c: e <= tol for c, e in per.items()}
        # a candidate that leaves the tolerance against the REFERENCE must do so at or after a look-ahead step in which the
        # reference's own edge selection hung on a near-tie (margin recorded beside its outputs by the generator), and stay
        # below the post-flip bound: two correct fp32 implementations then follow different, equally valid graphs
        ref_flips = []
        for c in ref_ids:
            if ref_within[c]:
                continue
            mg = ref[c][1]
            flipped = False                                  # an attributed flip happened at an earlier look-ahead step
            for h in range(want.shape[1]):
                if rerr[c][h] <= tol:
                    continue
                m = None if mg is None else float(np.minimum.accumulate(mg)[h])
                # what can flip a selection: a margin (in squared distance) below 4 * thr * (the deviation the two rollouts had
                # BEFORE this look-ahead step, at least fp32 accumulation noise of 1e-6) - not merely below what the stated
                # tolerance could cross.  After an attributed flip the later steps are its consequence.
                before = float(rerr[c][h - 1]) if h > 0 else 0.0
                reach = 4.0 * float(task["adj_thresh"]) * max(1e-6, min(before, tol))
                explained = rerr[c][h] <= post_flip_bound and (flipped or (m is not None and m < reach))
                flipped |= explained
                ref_unexplained |= not explained
                ref_flips.append({"candidate": int(c), "lookahead_step": int(h), "abs_err": float(rerr[c][h]),
                                  "reference_selection_margin": m, "margin_within_reach_of_the_deviation_before": reach,
                                  "near_tie_in_the_reference": bool(explained)})
        # candidates the older per-forward fixtures cover, once more through the oracle with the BLAS threading those fixtures'
        # pin was established with (8 threads: tests/test_fullsize_golden.py)
        o8_ids = [c for c in ref_ids if c in (0, 49, 487, 1023)]
        per8 = {}
        if o8_ids:
            o8, _, _, _ = _oracle_child(cloud, task, W, actions[o8_ids], 1, 8)
            per8 = {c: float(np.abs(gpu_seqs[picks.index(c)] - o8[i]).max()) for i, c in enumerate(o8_ids)}
        n_in = sum(1 for v in ref_within.values() if v)
        # how many candidates MAY leave the tolerance: those for which the reference's own record shows an edge selection hanging
        # on a near-tie somewhere in the rollout (nothing else is ever excused; no flat percentage)
        # (candidates whose record holds a margin below 4 * thr * 1e-6 - within reach of accumulation noise alone)
        tie_prone_ref = [c for c in ref_ids if ref[c][1] is not None and float(np.min(ref[c][1])) < 4.0 * float(task["adj_thresh"]) * 1e-6]
        n_out = len(ref_ids) - n_in
        ref_ok = bool(not ref_unexplained and n_out <= len(tie_prone_ref))
        clause = ("every covered candidate within tol at every step" if n_out == 0 else
                  f"{n_in} of {len(ref_ids)} within tol at every step; the other {n_out} leave it at / after a look-ahead step where the "
                  f"reference's own recorded selection margin is within reach of the deviation the rollouts had before it (4 x thr x max(1e-6, "
                  f"that deviation); {len(tie_prone_ref)} of the candidates hold a margin below 4 x thr x 1e-6 somewhere) and stay below the post-flip bound" if ref_ok else "FAILED: a deviation without a near-tie on record")
        vs_ref = {"candidates": ref_ids, "n_candidates": len(ref_ids), "candidates_within_tol_all_steps": n_in,
                  "max_abs_err": max(per.values()), "max_abs_err_within_tol": max([e for e in per.values() if e <= tol], default=None),
                  "per_candidate_max_abs_err": {str(c): e for c, e in per.items()},
                  "within_tol": bool(all(ref_within.values())), "tol": tol, "flips_vs_reference": ref_flips,
                  # every covered candidate is within tol of the reference over all steps, OR leaves it only at / after a look-ahead
                  # step in which the REFERENCE's own edge selection hung on a near-tie (its recorded margin < 4*adj_thresh*tol)
                  "within_tol_or_near_tie_in_the_reference": bool(not ref_unexplained),
                  "candidates_with_a_near_tie_on_record": len(tie_prone_ref), "ok": ref_ok, "ok_clause": clause,
                  "oracle_8_blas_threads_max_abs_err": {str(c): e for c, e in per8.items()},
                  "source": "tests/golden/full_cloth_seqs.npz (+ full_cloth_{a,flip}.npz): state_seqs the imported reference produced "
                            "for these candidates (tests/golden/make_golden.py --fullsize-r05 / --fullsize); start state, weights "
                            "(SHA-256), task scalars and raw actions checked bit-equal to this run's; free-running over all steps; "
                            "within_tol = every covered candidate <= tol with no tie attribution; a candidate beyond tol is listed "
                            "in flips_vs_reference with the REFERENCE's own selection margin at or before that look-ahead step"}
    flips = []
    for i in range(len(picks)):
        for h in range(err.shape[1]):
            if within[i, h]:
                continue
            c = int(picks[i])
            flips.append({"candidate": c, "lookahead_step": int(h), "abs_err": float(err[i, h]),
                          "oracle_selection_margin": float(np.minimum.accumulate(margin, axis=1)[i, h]),
                          # the GPU agrees with the REFERENCE on this candidate: the flip is the checker's (the single-thread
                          # oracle sums in another order than the reference and parted from both at the tie); false: the GPU
                          # parts from the reference too (listed in vs_reference.flips_vs_reference); null: no reference record
                          "checker_induced": (bool(ref_within[c]) if c in ref_within else None)})
    base = {"value": steps / dt, "unit": "rollout-steps/s", "cores": workers, "kind": "port",
            "sample": f"numpy oracle, {len(picks)} candidates of the timed batch x {steps // len(picks)} rollout steps "
                      f"(cloth 2025+1 particles), one candidate per process and BLAS thread, {dt:.1f}s wall"}
    tie_prone_oracle = int((np.min(margin, axis=1) < tie_margin).sum())
    ok = bool(not unexplained.any() and int((~clean).sum()) <= tie_prone_oracle and (vs_ref is None or vs_ref["ok"]))
    parity = {"candidates": [int(p) for p in picks], "max_abs_err": float(err[within].max()) if within.any() else None,
              "tol": tol, "candidates_within_tol_all_steps": int(clean.sum()), "edge_flips": flips,
              "edge_flips_checker_induced": sum(1 for f in flips if f["checker_induced"]),
              "edge_flips_gpu_vs_reference": sum(1 for f in flips if f["checker_induced"] is False),
              "vs_reference": vs_ref, "ok": ok,
              "ok_clause": ("all checked candidates within tol of the oracle" if clean.all() else
                            f"{int(clean.sum())} of {len(picks)} within tol of the oracle at every step; the others follow a near-tie in the "
                            f"oracle's own edge selection ({tie_prone_oracle} candidates have one)") + ("" if vs_ref is None else "; vs the reference: " + vs_ref["ok_clause"]),
              "what": "state_seqs of these candidates from the LAST TIMED step. vs_reference: against the reference's own "
                      "outputs for the candidates a committed fixture covers (r05: all of them; max-abs over all steps <= tol, or "
                      "a near-tie in the reference's own edge selection). Also, as the timed CPU leg, "
                      "against the oracle, free-running over all steps; max_abs_err is over the (candidate, look-ahead step) pairs "
                      "within tol; edge_flips lists the others, each of which must follow a near-tie in the oracle's own edge "
                      f"selection (margin < {tie_margin:.1e} in squared distance) at or before that look-ahead step and stay below "
                      f"{post_flip_bound:.0e} - otherwise ok is false; no more candidates may be excused than have such a near-tie "
                      "on record (ok_clause says which clause held). checker_induced = the GPU is within tol of the REFERENCE on that candidate, i.e. it is the "
                      "single-BLAS-thread oracle of this leg that parted at the tie (DESIGN.md section 4)"}
    return base, parity


def plumbing_only(args, world, rank, local_rank):
    """--plumbing-only: what the launch gave every rank, gathered over gloo and printed by rank 0 as one JSON line.  No GPU, no
    engine import: the CPU check of launch_ranks_if_needed (tests/test_bench_launch.py)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "world_env": world, "world_dist": dist.get_world_size(),
                                      "pid": os.getpid(), "argv": sys.argv[1:]})
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = [{"rank": rank, "local_rank": local_rank, "world_env": world, "world_dist": 1, "pid": os.getpid(), "argv": sys.argv[1:]}]
    if args.plumbing_only.startswith("fail-rank-") and rank == int(args.plumbing_only.rsplit("-", 1)[1]):
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"plumbing": seen, "n_gpus": world, "launched_by": os.environ.get("AG_BENCH_LAUNCHED_BY", "external launcher")}),
              flush=True)


def main():
    args = _parser().parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the launcher decides how many ranks exist; the line reports what dist sees (n_gpus = world size), never the flag
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: reporting n_gpus={world}",
              file=sys.stderr, flush=True)
    if args.plumbing_only:
        return plumbing_only(args, world, rank, local_rank)
    n_dev = torch.cuda.device_count()                       # (counting devices does not initialise HIP)
    if n_dev == 0:
        sys.exit(f"bench.py: rank {rank} sees no GPU")
    # Rehearsal hooks (never set by the driver): AG_BENCH_SHARE_GPU=1 maps every rank onto the GPUs that exist (two
    # ranks on a one-GPU box) and AG_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.
    if local_rank >= n_dev:
        # Either a launcher that isolates the GPUs per rank (every rank sees only its own device, as index 0) or fewer GPUs than
        # ranks.  Take the device that exists; with the nccl backend RCCL itself refuses two ranks on one physical device, loudly,
        # and multi_gpu.distinct_devices in the line says how many physical devices the ranks really ran on.
        if os.environ.get("AG_BENCH_SHARE_GPU") != "1":
            print(f"bench.py: rank {rank}: LOCAL_RANK {local_rank} but {n_dev} visible device(s): using device {local_rank % n_dev}",
                  file=sys.stderr, flush=True)
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # AG_BENCH_FORCE_DIST=1 (never set by the driver): also a ONE-rank run initialises the process group and takes the
    # collective path - all-gather of the rewards and both MAX all-reduces on RCCL with a world of one - so that the RCCL
    # bring-up (communicator on device_id, the calls themselves) executes on whatever hardware exists
    dist_on = world > 1 or os.environ.get("AG_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        backend = os.environ.get("AG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        import adaptigraph_amd.sharding as _sh
        _sh.FORCE_COLLECTIVES = world == 1                  # a world of one still issues its all-gather

    import adaptigraph_amd as ag

    rng = np.random.default_rng(0)
    cloud = cloth_cloud(args.side, rng)
    N_o = cloud.shape[0]
    B, H, R = args.candidates, args.lookahead, args.repeat
    task = make_task(max_nR=int(1.2 * 6 * (N_o + 1)) + 64, limits=not args.host_decode and args.repeat <= 10)
    Wt = random_weights(0)
    model = ag.DynamicsPredictor(*model_cfg(), dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in Wt.items()})
    ppm = types.SimpleNamespace(task_config=task, eef_num=1, material="cloth", material_dims=task["material_dims"],
                                material_indices=task["material_indices"],
                                physics_param={"cloth": torch.tensor([0.5])}, adj_thresh=task["adj_thresh"])
    from adaptigraph_amd.sharding import shard_bounds, all_gather_costs
    actions = torch.from_numpy(make_actions(B, H, R, cloud, rng))      # every rank draws the same full batch
    lo, hi = shard_bounds(B, world, rank)                              # contiguous shard
    a_local = actions[lo:hi].to(dev)
    state0 = torch.from_numpy(cloud).to(dev)
    # MPC objective of the cloth task (plan.py:139-174): chamfer to a target cloud + cloth collision penalty + bbox
    from functools import partial
    tgt = torch.from_numpy((cloud + np.float32([0.9, 0.0, 0.6]) + rng.normal(0, 0.02, cloud.shape)).astype(np.float32)).to(dev)
    bbox = np.array([[-0.45, 0.0], [-0.25, 0.45]]) * task["sim_real_ratio"] * 4.0
    err_fn = partial(ag.chamfer, y=tgt[None])
    pen_fn = partial(ag.cloth_penalty, sim_real_ratio=float(task["sim_real_ratio"]), group=True if dist_on else None)
    eng = model.engine(dev)
    if args.chunk:
        eng.set_chunk(args.chunk)
    flag = torch.zeros(64, dtype=torch.int32, device=dev)

    last = {}
    from adaptigraph_amd.sharding import sharded_candidate_rewards
    actions_dev = actions.to(dev)

    def rollout(a):
        seq = ag.dynamics(state0, a, model, dev, ppm, _sync=False, _overflow_flag=flag)["state_seqs"]   # (b, H, N_o, 3)
        last["seq"] = seq                                              # a reference, not a copy (parity_check below)
        return seq

    def reward(seq, a):
        # running_cost (plan.py:27-59): the two batch-global maxima are all-reduced (MAX) when the batch is sharded
        return ag.running_cost(seq, a, state0, error_func=err_fn, penalty_func=pen_fn, bbox=bbox,
                               group=True if dist_on else None)["reward_seqs"]

    def one_step():
        # shard -> rollout -> rewards -> all-gather (RCCL over xGMI: B/N fp32 per rank); tests/test_sharding_gloo.py
        # drives the same function with two gloo ranks
        return sharded_candidate_rewards(actions_dev, rollout, reward)

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync_all()
    # ---- timed region: EXACTLY args.steps rollouts, no per-kernel instrumentation (the engine overlaps alternate
    # candidate chunks on two in-library HIP streams, which per-kernel events would serialise)
    eng.set_profiling([])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        costs = one_step()
    sync_all()
    dt = time.perf_counter() - t0
    assert int(flag[0].item()) <= task["max_nR"], "a graph exceeded max_nR during the bench"
    assert int(flag[1].item()) <= args.repeat, "an action_repeat exceeded the task config's bound during the bench"
    assert torch.isfinite(costs).all()
    # candidates of the timed batch that the CPU leg re-computes with the oracle: first, last and evenly spaced ones
    # (every launch chunk of both streams is hit); their GPU results come from the LAST TIMED step
    # four candidates per worker core (16-core CPU share of a one-GPU box): ~12 s of CPU work
    n_pick = 0 if args.no_cpu_baseline or dist_on else max(2, min(4 * min(16, os.cpu_count() or 2), hi - lo))
    picks = parity_picks(hi - lo, n_pick)
    ref = reference_golden(cloud, task, Wt, actions.numpy(), with_margin=True) if picks else {}
    picks = sorted(set(picks) | {c for c in ref if lo <= c < hi})       # world == 1 here: candidate id == local index
    timed_seqs = last["seq"][picks].cpu().numpy() if picks else None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    multi = None
    if dist_on:
        # diagnostics for a multi-GPU run (not part of `value`): every rank's own wall time of the timed region, and the
        # latency of the step's exchange alone (MAX all-reduce of two scalars + all-gather of the B/N rewards)
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, tmax.clone())
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        probe = torch.zeros(hi - lo, device=dev)
        two = torch.zeros(2, device=dev)
        from adaptigraph_amd.sharding import all_gather_costs as _agc
        for _ in range(3):
            dist.all_reduce(two, op=dist.ReduceOp.MAX); _agc(probe, B)
        sync_all()
        tc = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(two, op=dist.ReduceOp.MAX); _agc(probe, B)
        sync_all()
        # which physical device every rank ran on - as plain integers through the tensor collectives the timed path itself uses (no
        # pickled objects: nothing in this diagnostic block may be the first thing to fail on a real multi-GPU run)
        props = torch.cuda.get_device_properties(dev)
        uuid = str(getattr(props, "uuid", "")) or ""
        import zlib
        mine = torch.tensor([rank, int(os.environ.get("LOCAL_RANK", "0")), dev.index if dev.index is not None else 0,
                             int(getattr(props, "pci_bus_id", -1) or -1), zlib.crc32(uuid.encode()), os.getpid()], device=dev, dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ids = [dict(zip(("rank", "local_rank", "device_index", "pci_bus_id", "uuid_crc32", "pid"), [int(v) for v in t.tolist()])) for t in allr]
        multi = {"world_size": dist.get_world_size(), "launched_by": os.environ.get("AG_BENCH_LAUNCHED_BY", "external launcher"),
                 "ranks": ids, "distinct_devices": len({(i["pci_bus_id"], i["uuid_crc32"]) if i["pci_bus_id"] >= 0 or i["uuid_crc32"] else
                                                        ("index", i["device_index"]) for i in ids}),
                 "per_rank_ms_per_step": [float(t.item()) / args.steps * 1e3 for t in per_rank],
                 "exchange_us_per_step": (time.perf_counter() - tc) / 20 * 1e6,
                 "backend": dist.get_backend(), "candidates_per_rank": [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]}
    dt = float(tmax.item())
    # ---- roofline pass: the same rollout once more with HIP events around every launch of the profiled kernels, on
    # the stream they are launched on.  Profiling pins the engine to ONE stream: with two chunks sharing the GPU an
    # event-bracketed duration measures the neighbour's kernels too.
    fams = [] if args.no_kernel_profile else ["edge_enc", "node_prop"] if not args.profile_all else [
        "edge_count", "edge_emit", "node_enc", "edge_enc", "node_prop", "node_final", "roll_init", "roll_update", "cost"]
    prof_steps = 1
    eng.reset_stats()
    if fams:
        # share_first off for this pass: every k_edge_enc launch is then a full 128-candidate one (with the default the first
        # forward of every chunk encodes the tool edges only and one tiny launch encodes the shared base graph, which would
        # mix three launch shapes into one average); the kernels themselves are the same
        with eng.options(share_first=0):
            eng.set_profiling(fams)
            for _ in range(prof_steps):
                one_step()
            sync_all()
            eng.set_profiling([])
    # the same kernel as it runs INSIDE a normal two-stream rollout (what rocprofv3 of the plain command averages):
    # events on both streams, durations include the other stream's co-running kernels
    ms_edge_co, n_edge_co = 0.0, 0
    if fams:
        fam_single = {f: eng.kernel_stats(f) for f in fams}
        eng.reset_stats()
        with eng.options(share_first=0):
            eng.set_profiling(["edge_enc"], keep_streams=True)
            one_step()
            sync_all()
            eng.set_profiling([])
        ms_edge_co, n_edge_co = eng.kernel_stats("edge_enc")
        eng.reset_stats()
    one_step()
    sync_all()
    share_counts = eng.share_counts()                       # (base edges, slots served by the shared table, slots encoded per candidate)
    fwd_executed, fwd_needed = eng.rollout_counts()          # candidate-forwards of that call (this rank's shard)
    # ---- secondary figure: the opt-in bf16x3 arithmetic (3-way bf16 split on the bf16 matrix pipe, fp32 accumulate;
    # validated at the same 1e-5 parity bar, tests/test_gpu_more.py).  NOT the headline: `value` is exact fp32.
    dt_b3 = None
    if not args.no_bf16x3:
        model.set_precision("bf16x3")
        for _ in range(args.warmup):
            one_step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            costs_b3 = one_step()
        sync_all()
        dt_b3 = time.perf_counter() - t0
        b3_fam = {}
        if fams:                                            # the bf16x3 kernels against THEIR peak (secondary object only)
            eng.reset_stats()
            with eng.options(share_first=0):
                eng.set_profiling(["edge_enc", "node_prop"])
                one_step()
                sync_all()
                eng.set_profiling([])
            b3_fam = {f: eng.kernel_stats(f) for f in ("edge_enc", "node_prop")}
            eng.reset_stats()
        model.set_precision("fp32")
        tb3 = torch.tensor([dt_b3], device=dev, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(tb3, op=dist.ReduceOp.MAX)
        dt_b3 = float(tb3.item())
        assert torch.isfinite(costs_b3).all()
    # ---- ms / MPC iteration (BASELINE metric, second half): sample -> rollout -> running_cost -> MPPI update -> best
    # candidate rolled out again (planner.py:234-277), whole batch in one call, exact fp32
    lo_lim = torch.tensor([float(cloud[:, 0].min()) - 0.5, float(cloud[:, 2].min()) - 0.5, -3.14, R + 0.5], device=dev)
    hi_lim = torch.tensor([float(cloud[:, 0].max()) + 0.5, float(cloud[:, 2].max()) + 0.5, 3.14, R + 0.5], device=dev)
    roll_fn = lambda s, a: ag.dynamics(s, a, model, dev, ppm, _sync=False, _overflow_flag=flag)
    eval_fn = partial(ag.running_cost, error_func=err_fn, penalty_func=pen_fn, bbox=bbox, group=True if dist_on else None)
    act0 = actions[0].to(dev)
    torch.manual_seed(1234)                                        # identical samples on every rank
    ms_mpc = None
    if not args.no_mpc_iter:
        # (reuse_best_rollout: the best candidate's rollout is taken out of the batch instead of being re-rolled with a batch of
        # one - exact on this engine, and what Planner does by default for the engine's own dynamics(); single rank only)
        ag.mpc_iteration(state0, act0, roll_fn, eval_fn, lo_lim, hi_lim, B, dev, group=True if dist_on else None, reuse_best_rollout=True)
        sync_all()
        t0 = time.perf_counter()
        n_mpc = 2
        for _ in range(n_mpc):
            mpc = ag.mpc_iteration(state0, act0, roll_fn, eval_fn, lo_lim, hi_lim, B, dev, group=True if dist_on else None,
                                   reuse_best_rollout=True)
        sync_all()
        tm = torch.tensor([(time.perf_counter() - t0) / n_mpc], device=dev, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ms_mpc = float(tm.item()) * 1e3
        assert torch.isfinite(mpc["reward_seqs"]).all()
    fam_ms = fam_single if fams else {}
    ms_edge, n_edge = fam_ms.get("edge_enc", (0.0, 0))
    if rank == 0:
        # edges per graph: measured on the start graph of candidate 0 (constant to within a few edges over the rollout)
        mask = torch.ones((1, N_o + 1), dtype=torch.bool, device=dev)
        tool = torch.zeros((1, N_o + 1), dtype=torch.bool, device=dev)
        tool[:, N_o:] = True
        pos = torch.cat([state0, torch.tensor([[float(a_local[0, 0, 0]), float(state0[:, 1].min()) + 0.1,
                                                float(a_local[0, 0, 1])]], device=dev)])[None]
        el0 = ag.construct_edges_index(pos, task["adj_thresh"], mask, tool, task["topk"], True)
        E = int(el0.n_edges[0])
        # self-loop edges are not run through the relation encoder (their C row is a model constant): count the FLOPs
        # of the edges the kernel actually encodes
        E_enc = E - int((el0.recv[0, :E] == el0.send[0, :E]).sum())
        total_steps = B * H * R
        launches_per_step = max(1, n_edge // prof_steps)
        edges_per_launch = E_enc * (hi - lo) * H * R / launches_per_step
        avg_ms = ms_edge / max(1, n_edge)
        def pmc_traffic(name, key, want):
            """PMC bytes per launch, measured off-line with rocprofv3 --pmc (profiles/README.md); null when the committed
            measurement was taken at another launch shape"""
            tpath = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(tpath):
                return None
            tj = json.load(open(tpath))
            if key not in tj or "hbm_bytes_per_launch" not in tj:
                return None
            return tj["hbm_bytes_per_launch"] if abs(tj[key] - want) <= 0.01 * want else None

        def latest(stem):                                   # the newest round's committed measurement of that name
            for r in ("r06", "r05", "r04", "r03"):
                if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_{stem}")):
                    return f"{r}_{stem}"
            return f"r06_{stem}"
        traffic_file = latest("traffic_k_edge_enc.json")
        traffic = pmc_traffic(traffic_file, "edges_per_launch", edges_per_launch)
        achieved = FLOP_PER_EDGE * edges_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # second-largest family: the propagate chain with the message passing fused into it (k_node_prop<false>, two
        # launches per rollout step).  It is bounded by BOTH resources, so both fractions are reported: the matrix work
        # (135,000 FLOP per particle) against the fp32 MFMA peak, and the compulsory HBM bytes against 8 TB/s - per
        # candidate one 640-B C row per encoded edge and a 4-B index per edge, streamed once; the previous round's U, V
        # and eff tables in, the new eff, U, V out (640 B per particle each; the per-edge V gathers re-read the V table
        # out of L2 / Infinity Cache).
        ms_np, n_np = fam_ms.get("node_prop", (0.0, 0))
        cand_per_launch = (hi - lo) * H * R * 2 / max(1, n_np // prof_steps)
        np_flop_launch = FLOP_PER_NODE_PROP * (N_o + 1) * cand_per_launch
        np_bytes_launch = (E_enc * 640 + E * 4 + (N_o + 1) * 640 * 6) * cand_per_launch
        np_avg_ms = ms_np / max(1, n_np)
        np_tflops = np_flop_launch / (np_avg_ms * 1e-3) / 1e12 if np_avg_ms > 0 else 0.0
        np_gbs = np_bytes_launch / (np_avg_ms * 1e-3) / 1e9 if np_avg_ms > 0 else 0.0
        np_traffic_file = latest("traffic_k_node_prop.json")
        np_traffic = pmc_traffic(np_traffic_file, "candidates_per_launch", cand_per_launch)
        line = {
            "metric": "rollout-steps/sec", "value": total_steps * args.steps / dt, "unit": "rollout-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: cloth, 1024 candidates x 20 rollout steps (2 look-ahead x "
                                   "repeat 10) x 2025+1 particles, radius graph rebuilt every step",
                       "candidates": B, "horizon": H * R, "particles": N_o + 1, "edges_per_graph": E, "edges_encoded_per_graph": E_enc,
                       "parallelism": f"candidates sharded over {world} GPU(s), all-gather of costs",
                       "ms_per_mpc_rollout": dt / args.steps * 1e3, "ms_per_mpc_iter": ms_mpc,
                       "action_path": "host decode (ag_rollout)" if "action_upper_lim" not in task else
                                      "device-planned (ag_rollout_actions: decode + launch plan on the GPU)"},
            "roofline": {"bound": "mfma", "kernel": "k_edge_enc", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_source": f"profiles/{traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch "
                                           "shape, committed - NOT measured by this run (PMC needs rocprofv3); null if the "
                                           "committed file was taken at another launch shape",
                         "avg_launch_ms": avg_ms, "launches": int(n_edge),
                         "measured": "HIP events on the launch stream, 1 extra rollout after the timed region with the "
                                     "engine pinned to one stream and share_first off (every launch a full 128-candidate one)",
                         "flop_per_edge": FLOP_PER_EDGE, "edges_per_launch": edges_per_launch,
                         "co_running": {"avg_launch_ms": ms_edge_co / max(1, n_edge_co), "launches": int(n_edge_co),
                                        "note": "same kernel inside a normal two-stream rollout: the duration spans "
                                                "whatever the other stream ran beside it (cf. the kernel trace of the "
                                                "plain command); not a roofline measure"}},
            "roofline_second_kernel": {"kernel": "k_node_prop<false> (propagate chain + fused message passing)",
                                       "avg_launch_ms": np_avg_ms, "launches": int(n_np),
                                       "mfma": {"achieved": np_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                "frac": np_tflops / PEAK_FP32_MFMA_TFLOPS, "flop_per_launch": np_flop_launch},
                                       "hbm": {"achieved": np_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": np_gbs / PEAK_HBM_GBS, "bytes_per_launch": np_bytes_launch,
                                               "traffic": np_traffic, "traffic_source": f"profiles/{np_traffic_file} (committed PMC passes)"},
                                       "note": "the gather of one workgroup (HBM / L2 bound) runs beside the matrix work of "
                                               "the other workgroup on its CU; achieved = algorithmic FLOPs resp. compulsory "
                                               "HBM bytes / HIP-event time; traffic = PMC FETCH_SIZE x2 + WRITE_SIZE per launch"},
            "kernel_ms_per_rollout_single_stream": {f: v[0] / prof_steps for f, v in fam_ms.items()},
            # SHA-256 of the gathered per-candidate reward vector of the last timed step (fp32 bytes): sharded == unsharded
            # bit for bit (tests/test_gpu_two_ranks.py compares the 2-rank line with the 1-rank line)
            "reward_sha256": __import__("hashlib").sha256(costs.detach().cpu().numpy().tobytes()).hexdigest(),
            # HIP multiplexes streams onto this many hardware queues; the engine's four + RCCL's need more than the default 4
            "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "hw_queues": ag.hw_queues},
        }
        # The reward vector of this batch is deterministic (no float atomics; sharded == unsharded bit for bit): a run of the default
        # workload - on any number of GPUs - must reproduce the SHA-256 of the committed one-GPU evidence line, whose parity_check
        # compared the same rollouts with the reference's own outputs.  This is the parity statement of an N > 1 line, which has
        # no CPU leg of its own.
        prof = os.path.join(ROOT, "profiles", latest("bench_default.json"))
        if os.path.exists(prof):
            pj = json.load(open(prof))
            same_workload = all(pj["config"].get(k) == line["config"].get(k) for k in ("candidates", "horizon", "particles", "edges_per_graph",
                                                                                        "action_path"))
            if same_workload and "reward_sha256" in pj:
                line["parity_vs_one_gpu_evidence"] = {"equal": pj["reward_sha256"] == line["reward_sha256"], "expected_sha256": pj["reward_sha256"],
                                                      "source": f"profiles/{latest('bench_default.json')} (its parity_check.vs_reference: "
                                                                f"{pj['parity_check']['vs_reference']['candidates_within_tol_all_steps']} of "
                                                                f"{pj['parity_check']['vs_reference']['n_candidates']} candidates within 1e-5 of the reference)"}
        if multi is not None:
            line["multi_gpu"] = multi
        # whole-rollout arithmetic rate: FLOPs the kernels execute per rollout step and candidate
        # (encoded edges x 140,100 + particles x (2 x 135,000 + 135,900) + 8 N^2 for the graph), and the reference
        # formulation's F_ref = N*361,800 + N_o*90,900 + E*500,100 + 8 N^2 over the same time ("effective")
        Np = N_o + 1
        f_exec = E_enc * FLOP_PER_EDGE + Np * (2 * 135000 + 135900) + 8 * Np * Np
        f_ref = Np * 361800 + N_o * 90900 + E * 500100 + 8 * Np * Np
        t_step = dt / args.steps
        # FLOPs one timed call EXECUTES (rank 0's shard scaled to the whole batch): f_exec per candidate-forward that ran,
        # minus the edges of the first forward that the shared table served instead of the relation encoder (share_first:
        # `slots served` edges were NOT encoded, the base graph's were, once)
        shard = (hi - lo) / B
        fwd_exec_all = fwd_executed / shard
        saved_edges = (share_counts[1] - share_counts[0]) / shard if share_counts[0] else 0.0
        flop_call = f_exec * fwd_exec_all - saved_edges * FLOP_PER_EDGE
        line["end_to_end"] = {"executed_tflops": flop_call / t_step / 1e12,
                              "frac_of_fp32_mfma_peak": flop_call / t_step / 1e12 / PEAK_FP32_MFMA_TFLOPS / world,
                              "effective_reference_formulation_tflops": f_ref * total_steps / t_step / 1e12,
                              "flop_per_step_executed": f_exec, "flop_per_step_reference_formulation": f_ref,
                              "flop_per_call_executed": flop_call,
                              "candidate_forwards_executed": int(round(fwd_exec_all)), "candidate_forwards_needed": int(round(fwd_needed / shard)),
                              "edges_not_encoded_thanks_to_the_shared_first_forward": int(round(saved_edges)),
                              "how": "flop_per_call_executed = flop_per_step_executed x candidate_forwards_executed - "
                                     "edges_not_encoded x flop_per_edge (ag_ctx_rollout_counts / ag_ctx_share_counts of one call of the "
                                     "timed shape); executed_tflops = that / ms_per_step"}
        # the first forward of the timed call: edges the relation encoder ran over with / without the shared base table
        line["shared_first_forward"] = {"base_edges_encoded_once": share_counts[0], "slots_served_by_the_shared_table": share_counts[1],
                                        "slots_encoded_per_candidate": share_counts[2],
                                        "edges_encoded_without_sharing": share_counts[1] + share_counts[2]}
        if dt_b3 is not None:
            line["bf16x3_mode"] = {"value": total_steps * args.steps / dt_b3, "unit": "rollout-steps/s",
                                   "ms_per_step": dt_b3 / args.steps * 1e3, "dtype": "bf16x3 split, f32 accumulate",
                                   "note": "opt-in ag_ctx_set_precision(1); same 1e-5 parity bar; not the headline"}
            if b3_fam.get("edge_enc", (0, 0))[1]:
                # six bf16 partial products per fp32 product: executed bf16-MFMA FLOPs = 6 x the algorithmic fp32 FLOPs
                b3_ms = b3_fam["edge_enc"][0] / b3_fam["edge_enc"][1]
                b3_tf = 6 * FLOP_PER_EDGE * (E_enc * (hi - lo) * H * R / b3_fam["edge_enc"][1]) / (b3_ms * 1e-3) / 1e12
                np3_ms = b3_fam["node_prop"][0] / max(1, b3_fam["node_prop"][1])
                np3_tf = 6 * np_flop_launch / (np3_ms * 1e-3) / 1e12 if np3_ms > 0 else 0.0
                line["bf16x3_mode"]["roofline"] = {
                    "bound": "mfma", "kernel": "k_edge_enc_b3", "achieved": b3_tf, "peak": PEAK_BF16_MFMA_TFLOPS,
                    "unit": "TFLOP/s of bf16 MFMA (6 partial products per fp32 product)", "frac": b3_tf / PEAK_BF16_MFMA_TFLOPS,
                    "avg_launch_ms": b3_ms, "launches": int(b3_fam["edge_enc"][1]),
                    "fp32_equivalent_tflops": b3_tf / 6,
                    "second_kernel": {"kernel": "k_node_prop_b3<false>", "avg_launch_ms": np3_ms, "achieved": np3_tf,
                                      "frac": np3_tf / PEAK_BF16_MFMA_TFLOPS},
                    "note": "secondary object; PMC MFMA-busy: profiles/r05_bf16x3_*.json; what bounds it (weight delivery "
                            "through the LDS, not VALU issue): profiles/r05_bf16x3_limiter.json"}
        if not dist_on and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity_check"] = cpu_baseline(cloud, task, Wt, actions.numpy(), picks, timed_seqs, ref)
        print(json.dumps(line), flush=True)
        if "parity_check" in line and not line["parity_check"]["ok"]:
            sys.exit("bench: the timed rollout differs from the reference fixtures / the oracle by more than 1e-5")
        if line.get("parity_vs_one_gpu_evidence", {}).get("equal") is False:
            sys.exit("bench: the reward vector of the default batch differs from the committed one-GPU evidence run's (profiles/): "
                     "sharded and unsharded evaluations must agree bit for bit")
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
