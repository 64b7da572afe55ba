"""SURVEY §8(f) rank 2, host side: adaptigraph_amd.planner.Planner against vectors recorded from the reference's Planner
(src/planning/real_world/planner.py:38-323) driven with the closed-form rollout / cost stand-ins of tests/helpers.py
(tests/golden/make_golden.py --planner).  Pure host logic: runs on CPU tensors, no engine involved.

Bit-exact: the class only samples (same torch calls in the same order), calls the two callables, and takes argmax /
softmax means with the same torch operations as the reference.
"""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import load_golden, toy_rollout, toy_cost, toy_planner_config
from adaptigraph_amd.planner import Planner, farthest_points


def _planner(**over):
    cfg = toy_planner_config(toy_rollout, toy_cost)
    cfg.update(over)
    pl = Planner(cfg)
    pl.sample_action_sequences = lambda a, iter_index=None: pl.sample_action_sequences_default(a)
    return pl


def test_mppi_loop_defaults_match_reference():
    g = load_golden("planner")
    pl = _planner()
    act0, state_cur = torch.from_numpy(g["act0"]), torch.from_numpy(g["state_cur"])
    torch.manual_seed(32)
    res = pl.trajectory_optimization(state_cur, act0.clone())
    assert np.array_equal(res["act_seq"].numpy(), g["a_act_seq"])
    assert np.array_equal(res["best_model_output"]["state_seqs"].numpy(), g["a_best_state"])
    assert np.array_equal(res["best_eval_output"]["reward_seqs"].numpy(), g["a_best_reward"])
    assert res["model_outputs"] is None and res["eval_outputs"] is None
    torch.manual_seed(33)
    assert np.array_equal(pl.sample_action_sequences_default(act0.clone()).numpy(), g["a_sample"])
    mean = pl.optimize_action_mppi_default(torch.from_numpy(g["a_sample"]).clone(), torch.from_numpy(g["a_rewards"]))
    assert np.array_equal(mean.numpy(), g["a_mppi_mean"])
    x = torch.tensor([[9.0, -9.0, 0.3]])
    assert pl.clip_actions_default(x) is x and x.tolist() == [[0.5, -0.4000000059604645, 0.30000001192092896]]


def test_verbose_keeps_every_iteration():
    g = load_golden("planner")
    pl = _planner(verbose=True, n_update_iter=2)
    torch.manual_seed(35)
    res = pl.trajectory_optimization(torch.from_numpy(g["state_cur"]), torch.from_numpy(g["act0"]).clone())
    assert np.array_equal(res["act_seq"].numpy(), g["b_act_seq"])
    assert len(res["model_outputs"]) == 2 and len(res["eval_outputs"]) == 2
    assert np.array_equal(np.stack([e["reward_seqs"].numpy() for e in res["eval_outputs"]]), g["b_rewards"])


def test_fps_sampler_matches_reference():
    g = load_golden("planner")
    cfg = toy_planner_config(toy_rollout, toy_cost, action_dim=2, noise_type="fps", n_sample=9, n_update_iter=1)
    cfg["action_lower_lim"], cfg["action_upper_lim"] = torch.tensor([0.0, -0.1]), torch.tensor([0.2, 0.1])
    pl = Planner(cfg)
    assert np.array_equal(pl.sample_action_sequences_default(torch.zeros(3, 2)).numpy(), g["c_fps"])
    pts = np.array([[0.0, 0, 1, 0], [0, 0, 0, 0], [5, 5, 5, 5], [0, 0, 3, 0]], np.float64)
    assert np.array_equal(farthest_points(pts, 2), pts[[3, 2]])             # starts at the largest half-to-half motion
    assert np.array_equal(farthest_points(pts, 2, init_idx=1), pts[[1, 2]])


def test_chunk_loop_and_merge_match_reference_and_chunked_entry_equals_loop():
    g = load_golden("planner")
    act0, state_cur = torch.from_numpy(g["act0"]), torch.from_numpy(g["state_cur"])
    pl = _planner(n_update_iter=1)
    torch.manual_seed(36)
    res_all = []
    for ci in range(5):
        pl.chunk_id = ci
        res_all.append(pl.trajectory_optimization(state_cur, act0.clone()))
    merged = pl.merge_res(res_all)
    assert np.array_equal(np.stack([r["act_seq"].numpy() for r in res_all]), g["d_chunk_act_seqs"])
    assert np.array_equal(np.array([r["best_eval_output"]["reward_seqs"].mean().item() for r in res_all]), g["d_chunk_scores"])
    assert np.array_equal(merged["act_seq"].numpy(), g["d_act_seq"])
    assert np.array_equal(merged["best_eval_output"]["reward_seqs"].numpy(), g["d_best_reward"])
    end_state = torch.get_rng_state()
    # one rollout call for all chunks + one for the winners: same winner, same outputs, generator left in the same state
    calls = []
    pl2 = _planner(n_update_iter=1, model_rollout_fn=lambda s, a: (calls.append(a.shape[0]), toy_rollout(s, a))[1])
    torch.manual_seed(36)
    fused = pl2.trajectory_optimization_chunked(state_cur, act0.clone(), 5)
    assert calls == [5 * 16, 5]
    assert torch.equal(torch.get_rng_state(), end_state)
    assert np.array_equal(fused["act_seq"].numpy(), g["d_act_seq"])
    assert np.array_equal(fused["best_eval_output"]["reward_seqs"].numpy(), g["d_best_reward"])
    assert torch.equal(fused["best_model_output"]["state_seqs"], merged["best_model_output"]["state_seqs"])
    # configurations the fused entry does not cover fall back to the loop
    pl3 = _planner(n_update_iter=2)
    torch.manual_seed(5)
    a = pl3.trajectory_optimization_chunked(state_cur, act0.clone(), 2)
    torch.manual_seed(5)
    b = pl3.merge_res([pl3.trajectory_optimization(state_cur, act0.clone()) for _ in range(2)])
    assert torch.equal(a["act_seq"], b["act_seq"])


def test_reuse_of_the_best_candidates_in_batch_rollout_is_exact_for_a_batch_independent_rollout():
    """config['reuse_best_rollout'] (off by default): the best sequence's rollout is sliced out of its batch instead of being
    rolled out again with a batch of one - same result dictionaries for a rollout that does not depend on the batch (the
    closed-form stand-in here; the HIP engine on the GPU: tests/test_gpu_more.py), one rollout call less per chunk."""
    g = load_golden("planner")
    act0, state_cur = torch.from_numpy(g["act0"]), torch.from_numpy(g["state_cur"])
    out = {}
    for reuse in (False, True):
        calls = []
        pl = _planner(n_update_iter=2, reuse_best_rollout=reuse,
                      model_rollout_fn=lambda s, a: (calls.append(a.shape[0]), toy_rollout(s, a))[1])
        torch.manual_seed(35)
        res = pl.trajectory_optimization(state_cur, act0.clone())
        pl1 = _planner(n_update_iter=1, reuse_best_rollout=reuse,
                       model_rollout_fn=lambda s, a: (calls.append(-a.shape[0]), toy_rollout(s, a))[1])
        torch.manual_seed(36)
        ch = pl1.trajectory_optimization_chunked(state_cur, act0.clone(), 3)
        out[reuse] = (res, ch, list(calls))
    (r0, c0, k0), (r1, c1, k1) = out[False], out[True]
    assert k0 == [16, 16, 1, -48, -3] and k1 == [16, 16, -48]
    for a, b in ((r0, r1), (c0, c1)):
        assert torch.equal(a["act_seq"], b["act_seq"])
        assert torch.equal(a["best_model_output"]["state_seqs"], b["best_model_output"]["state_seqs"])
        assert a["best_model_output"]["state_seqs"].shape == b["best_model_output"]["state_seqs"].shape
        assert torch.equal(a["best_eval_output"]["reward_seqs"], b["best_eval_output"]["reward_seqs"])


def test_config_validation_and_gd():
    cfg = toy_planner_config(toy_rollout, toy_cost)
    bad = dict(cfg); bad.pop("n_sample")
    with pytest.raises(KeyError):
        Planner(bad)
    bad = dict(cfg); bad["planner_type"] = "CEM"
    with pytest.raises(AssertionError):
        Planner(bad)
    bad = dict(cfg); bad["action_lower_lim"] = torch.zeros(2)
    with pytest.raises(AssertionError):
        Planner(bad)
    gd = dict(cfg); gd["planner_type"] = "GD"
    with pytest.raises(NotImplementedError):
        Planner(gd).trajectory_optimization(torch.zeros(4, 3), torch.zeros(3, 3))
    with pytest.raises(AssertionError):
        Planner(cfg).trajectory_optimization(torch.zeros(4, 3), torch.zeros(2, 3))      # wrong horizon


# ------------------------------------------------------------------------------------------------- chunks over 2 ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _chunk_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("planner")
    act0, state_cur = torch.from_numpy(g["act0"]), torch.from_numpy(g["state_cur"])
    calls = []
    pl = _planner(n_update_iter=1, group=True, model_rollout_fn=lambda s, a: (calls.append(a.shape[0]), toy_rollout(s, a))[1])
    torch.manual_seed(36)
    res = pl.trajectory_optimization_chunked(state_cur, act0.clone(), 5)          # 3 + 2 chunks
    ok = np.array_equal(res["act_seq"].numpy(), g["d_act_seq"]) and \
        np.array_equal(res["best_eval_output"]["reward_seqs"].numpy(), g["d_best_reward"])
    # an evaluation that would all-reduce inside every call (built with group=..., as mpc_iteration wants it) is refused
    # BEFORE any rank enters it: the ranks evaluate 3 and 2 chunks, the collectives would not pair up (a hang)
    from functools import partial

    def cost_with_collective(state_seqs, act_seqs, state_cur=None, group=None, **kw):
        if group is not None:
            dist.all_reduce(torch.zeros(1))
        return toy_cost(state_seqs, act_seqs, state_cur=state_cur)
    bad = _planner(n_update_iter=1, group=True, evaluate_traj_fn=partial(cost_with_collective, group=True))
    try:
        bad.trajectory_optimization_chunked(state_cur, act0.clone(), 5)
        ok = False
    except ValueError as e:
        ok = ok and "rank-local" in str(e)
    nested = _planner(n_update_iter=1, group=True,
                      evaluate_traj_fn=partial(toy_cost, penalty_func=partial(cost_with_collective, group=True)))
    try:
        nested.trajectory_optimization_chunked(state_cur, act0.clone(), 5)
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok), calls))
    dist.destroy_process_group()


def test_chunks_dealt_to_two_gloo_ranks_give_the_reference_result():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, [3 * 16, 5]), (1, True, [2 * 16, 5])]


# ------------------------------------------- the reference's UNCHANGED chunk loop dealt to the ranks of config['group'] (r06)
def _reference_loop(pl, state_cur, act0, n_chunk):
    """plan.py:210, 241-247 as written there"""
    pl.total_chunks = n_chunk
    res_all = []
    for ci in range(n_chunk):
        pl.chunk_id = ci
        res = pl.trajectory_optimization(state_cur, act0)
        for k, v in res.items():
            res[k] = v.detach().clone() if isinstance(v, torch.Tensor) else v
        res_all.append(res)
    return pl.merge_res(res_all), res_all


def _loop_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("planner")
    act0, state_cur = torch.from_numpy(g["act0"]), torch.from_numpy(g["state_cur"])
    out = {"rank": rank}
    # (1) the golden case of the reference's own Planner: 5 chunks, n_update_iter 1, seed 36
    calls = []
    pl = _planner(n_update_iter=1, group=True, model_rollout_fn=lambda s, a: (calls.append(a.shape[0]), toy_rollout(s, a))[1])
    torch.manual_seed(36)
    merged, res_all = _reference_loop(pl, state_cur, act0.clone(), 5)
    out["golden"] = bool(np.array_equal(merged["act_seq"].numpy(), g["d_act_seq"]) and
                         np.array_equal(merged["best_eval_output"]["reward_seqs"].numpy(), g["d_best_reward"]))
    out["calls"] = list(calls)
    out["owners"] = [r["_chunk_owner"] for r in res_all]
    out["placeholders_are_nan"] = all(bool(torch.isnan(r["act_seq"]).all()) == (r["_chunk_owner"][1] != rank) for r in res_all)
    # (2) n_update_iter 3 (the in-call where() selection), 7 chunks, against the one-rank loop run in this very process
    one = _planner(n_update_iter=3)
    torch.manual_seed(77)
    want, _ = _reference_loop(one, state_cur, act0.clone(), 7)
    end_state = torch.get_rng_state()
    pl3 = _planner(n_update_iter=3, group=True)
    torch.manual_seed(77)
    got, _ = _reference_loop(pl3, state_cur, act0.clone(), 7)
    out["equal_one_rank"] = bool(torch.equal(got["act_seq"], want["act_seq"]) and
                                 torch.equal(got["best_model_output"]["state_seqs"], want["best_model_output"]["state_seqs"]) and
                                 torch.equal(got["best_model_output"]["action_seqs"], want["best_model_output"]["action_seqs"]) and
                                 torch.equal(got["best_eval_output"]["reward_seqs"], want["best_eval_output"]["reward_seqs"]) and
                                 sorted(got) == sorted(want))
    out["generator_in_step"] = bool(torch.equal(torch.get_rng_state(), end_state))
    # (3) a second series on the same planner works (the series counter was closed by merge_res)
    torch.manual_seed(77)
    again, _ = _reference_loop(pl3, state_cur, act0.clone(), 7)
    out["second_series"] = bool(torch.equal(again["act_seq"], want["act_seq"]))
    # (4) "Exceeds max dims" on ONE rank's call (call 2 of 6 -> rank 2 % world) is raised by merge_res on EVERY rank
    n_calls = [0]

    def overflowing(s, a):
        n_calls[0] += 1
        if a.shape[0] > 1 and float(a[0, 0, 0]) == float(marker[0]):
            raise Exception("Exceeds max dims")
        return toy_rollout(s, a)
    bad = _planner(n_update_iter=1, group=True, model_rollout_fn=overflowing)
    torch.manual_seed(5)
    probe = [_planner(n_update_iter=1).sample_action_sequences(act0.clone(), iter_index=0) for _ in range(3)]
    marker = [float(probe[2][0, 0, 0])]                           # first number of the samples of call 2
    torch.manual_seed(5)
    try:
        _reference_loop(bad, state_cur, act0.clone(), 6)
        out["error_everywhere"] = False
    except Exception as e:  # noqa: BLE001
        out["error_everywhere"] = str(e) == "Exceeds max dims"
    # ... and the planner is usable afterwards
    torch.manual_seed(36)
    merged2, _ = _reference_loop(pl, state_cur, act0.clone(), 5)
    out["usable_after_error"] = bool(np.array_equal(merged2["act_seq"].numpy(), g["d_act_seq"]))
    # (5) callables that issue their own collectives are refused on every rank (at the merge, like every error of the loop)
    from functools import partial

    def cost_with_collective(state_seqs, act_seqs, state_cur=None, group=None, **kw):
        return toy_cost(state_seqs, act_seqs, state_cur=state_cur)
    refused = _planner(n_update_iter=1, group=True, evaluate_traj_fn=partial(cost_with_collective, group=True))
    try:
        _reference_loop(refused, state_cur, act0.clone(), 4)
        out["refused"] = False
    except ValueError as e:
        out["refused"] = "rank-local" in str(e)
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_the_references_chunk_loop_is_dealt_to_the_ranks_and_merges_to_the_one_rank_result(world):
    """planner_config['group'] + `planner.total_chunks = n_chunk` (plan.py:210): call ci of the loop of plan.py:241-247 runs on
    rank ci % world, the others draw its samples and get a placeholder; merge_res all-gathers the winners and broadcasts the best
    one's outputs.  Bit-equal to the one-rank loop (and to the reference Planner's own recorded result), generators in step,
    errors raised on every rank.  world 8 (the node's size): fewer calls than ranks in every series here - ranks that own no call
    still draw, still take part in the merge."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, o in enumerate(res):
        mine = [k for k in range(5) if k % world == r]
        assert o["calls"] == [16, 1] * len(mine), o              # only its own chunks were rolled out (+ each one's winner, planner.py:270)
        assert o["owners"] == [(k, k % world) for k in range(5)]
        for key in ("golden", "placeholders_are_nan", "equal_one_rank", "generator_in_step", "second_series", "error_everywhere",
                    "usable_after_error", "refused"):
            assert o[key] is True, (r, key, o)


# ------------------------------------------------------------------------------- r06: when calls are dealt, and saying when not
def _engine_planner(upper_len=10.0, task_upper=10.0, **over):
    """a Planner configured like plan.py:177-207 around the ENGINE's dynamics() - constructed only, nothing runs (no GPU here)"""
    import types
    from functools import partial
    import adaptigraph_amd as ag
    mc = dict(verbose=False, nf_particle=150, nf_relation=150, nf_effect=150, nf_physics=10, attr_dim=2, state_dim=0, offset_dim=0,
              action_dim=3, density_dim=0, pstep=3, sequence_len=4, rel_particle_dim=0, rel_attr_dim=2, rel_group_dim=1,
              rel_distance_dim=3, rel_density_dim=0)
    m = ag.DynamicsPredictor(mc, {"material_index": {"rope": 0}, "rope": {"physics_params": [{"name": "p", "use": True}]}},
                             {"n_his": 4, "materials": ["rope"]}, "cuda:0")
    task = dict(adj_thresh=0.5, topk=10, connect_tools_all=False, sim_real_ratio=10, push_length=0.1, gripper_enable=False, max_n=1,
                max_nR=4000, n_his=4, eef_num=1, material="rope", pusher_points=[[0.0, 0.0, 0.12]], material_dims={"rope": 1},
                material_indices={"rope": 0})
    if task_upper is not None:
        task["action_upper_lim"] = [0.0, 4.5, 3.14, task_upper]
    ppm = types.SimpleNamespace(task_config=task, eef_num=1, material="rope", material_dims=task["material_dims"],
                                material_indices=task["material_indices"], physics_param={"rope": torch.tensor([0.5])}, adj_thresh=0.5)
    lo, hi = torch.tensor([-4.5, -2.5, -3.14, 2.0]), torch.tensor([0.0, 4.5, 3.14, upper_len])
    cfg = {"action_dim": 4, "model_rollout_fn": partial(ag.dynamics, model=m, device="cuda:0", ppm_optimizer=ppm),
           "evaluate_traj_fn": toy_cost, "n_sample": 8, "n_look_ahead": 1, "n_update_iter": 1, "reward_weight": 1.0,
           "sampling_action_seq_fn": partial(ag.sample_action_seq, action_lower_lim=lo, action_upper_lim=hi, n_sample=8, device="cuda:0"),
           "optimize_action_mppi_fn": partial(ag.optimize_action_mppi, action_lower_lim=lo, action_upper_lim=hi),
           "clip_action_seq_fn": partial(ag.clip_actions, action_lower_lim=lo, action_upper_lim=hi),
           "action_lower_lim": lo, "action_upper_lim": hi, "planner_type": "MPPI", "device": "cuda:0", "rollout_best": True}
    cfg.update(over)
    return Planner(cfg)


def test_dealing_needs_repeats_that_cannot_leave_the_task_configs_bound():
    """A call that does not wait for its rollout cannot fall back to the host decode for a push longer than the task config
    allows (forward_dynamics.py:156 accepts any length): calls are dealt only when the limits the sampler, the MPPI update and
    the clamp work with keep int(length) within task_config['action_upper_lim'][3]."""
    assert _engine_planner()._repeats_within_bound() is None
    assert _engine_planner(upper_len=10.9)._repeats_within_bound() is None          # int(10.9) = 10 <= 10
    why = _engine_planner(upper_len=12.0)._repeats_within_bound()
    assert why is not None and "12.0" in why and "10" in why
    assert "action_upper_lim" in _engine_planner(task_upper=None)._repeats_within_bound()
    own = _engine_planner(sampling_action_seq_fn=lambda a, iter_index=None: a[None].repeat(8, 1, 1))
    assert "sampling_action_seq_fn" in own._repeats_within_bound()
    # the planner's own defaults clamp to its own limits
    d = _engine_planner()
    d.sample_action_sequences, d.optimize_action_mppi, d.clip_action_sequences = (d.sample_action_sequences_default,
                                                                                  d.optimize_action_mppi_default, d.clip_actions_default)
    d._bound_ok = None
    assert d._repeats_within_bound() is None


def test_planner_says_once_why_calls_are_not_dealt(caplog):
    import logging
    import adaptigraph_amd as ag
    with caplog.at_level(logging.WARNING, logger="adaptigraph_amd.planner"):
        pl = _engine_planner()
        assert pl.pipeline_chunks == 6 and pl._pipeline_off_static() is None and not caplog.records      # nothing to say
        inner = pl.model_rollout
        _engine_planner(model_rollout_fn=lambda s, a: inner(s, a))                                         # a lambda around dynamics()
        assert len(caplog.records) == 1 and "functools.partial(adaptigraph_amd.dynamics" in caplog.records[0].getMessage()
        _engine_planner(verbose=True)
        _engine_planner(pipeline_chunks=0)
        msgs = [r.getMessage() for r in caplog.records]
        assert len(msgs) == 3 and "verbose" in msgs[1] and "pipeline_chunks" in msgs[2]
        assert all(m.startswith("planner: calls are not dealt to side streams") for m in msgs)
    # a planner on CPU stand-ins (the tests above) is not told anything
    with caplog.at_level(logging.WARNING, logger="adaptigraph_amd.planner"):
        caplog.clear()
        _planner()
        assert not caplog.records
