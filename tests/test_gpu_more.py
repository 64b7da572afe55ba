"""More GPU parity (-m gpu): path cross-checks, full-size configs, edge cases and error behaviour."""
import types

import numpy as np
import pytest
import torch

from test_gpu_parity import _cfg, _ppm, _edges_to_lists, POS_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ag():
    import adaptigraph_amd
    return adaptigraph_amd


@pytest.fixture(scope="module")
def O():
    from oracle import adaptigraph_oracle
    return adaptigraph_oracle


def _task(material, **kw):
    base = dict(sim_real_ratio=10, max_n=1, n_his=4, material=material, material_dims={material: 1},
                material_indices={material: 0})
    cfgs = {
        "rope": dict(adj_thresh=0.5, topk=10, connect_tools_all=False, push_length=0.1, gripper_enable=False,
                     eef_num=1, pusher_points=[[0.0, 0.0, 0.12]], max_nR=6000),
        "granular": dict(adj_thresh=0.4, topk=20, connect_tools_all=False, push_length=0.2, gripper_enable=False,
                         eef_num=5, max_nR=30000,
                         pusher_points=[[0, 0, 0.1], [0, 0.05, 0.1], [0, 0.025, 0.1], [0, -0.025, 0.1], [0, -0.05, 0.1]]),
        "cloth": dict(adj_thresh=0.75, topk=5, connect_tools_all=True, push_length=0.1, gripper_enable=True,
                      eef_num=1, pusher_points=[[0.0, 0.0, 0.17]], max_nR=16000),
    }
    base.update(cfgs[material])
    base.update(kw)
    return base


def _grid(side, pitch, jitter, rng):
    g = (np.arange(side) - (side - 1) / 2.0) * pitch
    xx, zz = np.meshgrid(g, g, indexing="ij")
    p = np.stack([xx.ravel() - 2.0, np.zeros(side * side), zz.ravel() + 1.0], 1)
    return (p + rng.normal(0, jitter, p.shape)).astype(np.float32)


def _rope(n, rng):
    t = np.linspace(0, 1, n)
    p = np.stack([-2 + 3 * t, 0 * t, 0.5 * np.sin(6 * t)], 1)
    return (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)


def _actions(cloud, B, H, rep, rng, spread=0.6):
    a = np.zeros((B, H, 4), np.float32)
    c = cloud.mean(0)
    a[..., 0] = c[0] + rng.uniform(-spread, spread, (B, H))
    a[..., 1] = c[2] + rng.uniform(-spread, spread, (B, H))
    a[..., 2] = rng.uniform(-3.1, 3.1, (B, H))
    rep = np.asarray(rep, np.float32)
    a[..., 3] = (rep[:, None] if rep.ndim == 1 else rep) + 0.5
    return a


def _model(ag, O, material, seed, dev, pstep=3):
    W = O.random_weights(seed)
    m = ag.DynamicsPredictor(*_cfg(material, pstep), dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return W, m


def test_rollout_step_equals_plain_forward_bitwise(ag, O, dev):
    """The rollout engine (class table, self-loop constants, on-device bookkeeping) and the plain ag_forward path
    (every row encoded, every edge encoded) must give the same bits for one step on the same graph."""
    rng = np.random.default_rng(3)
    task = _task("cloth")
    W, m = _model(ag, O, "cloth", 3, dev)
    cloud = _grid(14, 0.3, 0.02, rng)
    N_o, M = cloud.shape[0], 1
    B = 3
    a = _actions(cloud, B, 1, 1, rng)
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "cloth"))
    # the same step assembled by hand, reference-style
    dec, _ = O.decode_action(a, task["push_length"])
    xz, delta = O.tool_keypoints(dec, a[..., 2], task)
    state = np.zeros((B, 4, N_o + M, 3), np.float32)
    action = np.zeros((B, N_o + M, 3), np.float32)
    for b in range(B):
        y = np.float32(np.float32(cloud[:, 1].min()) + np.float32(0.01 * task["sim_real_ratio"]))
        tool = np.array([[xz[b, 0, 0, 0], y, xz[b, 0, 0, 1]]], np.float32)
        state[b, :] = np.concatenate([cloud, tool], 0)[None]
        action[b, N_o:] = delta[b, 0]
    attrs = np.zeros((B, N_o + M, 2), np.float32)
    attrs[:, :N_o, 0] = 1
    attrs[:, N_o:, 1] = 1
    mask = torch.ones((B, N_o + M), dtype=torch.bool, device=dev)
    toolm = torch.zeros((B, N_o + M), dtype=torch.bool, device=dev)
    toolm[:, N_o:] = True
    st = torch.from_numpy(state).to(dev)
    el = ag.construct_edges_index(st[:, -1], task["adj_thresh"], mask, toolm, task["topk"], True)
    pos, _ = m(state=st, attrs=torch.from_numpy(attrs).to(dev), edges=el,
               p_instance=torch.ones((B, N_o, 1), device=dev), action=torch.from_numpy(action).to(dev),
               cloth_physics_param=torch.full((B, 1), 0.5, device=dev))
    assert torch.equal(out["state_seqs"][:, 0], pos)


@pytest.mark.parametrize("material,cloud_fn,B,H,rep", [
    ("rope", lambda r: _rope(300, r), 64, 2, 3),                       # BASELINE configs[1] shape
    ("granular", lambda r: _grid(32, 0.12, 0.02, r), 6, 1, 3),         # configs[2] particle count, 5-point pusher
    ("cloth", lambda r: _grid(45, 0.3, 0.02, r), 5, 1, 2),             # configs[3] particle count
])
def test_full_size_configs_vs_oracle_and_sharding(ag, O, dev, material, cloud_fn, B, H, rep):
    rng = np.random.default_rng(17)
    task = _task(material)
    W, m = _model(ag, O, material, 17, dev)
    cloud = cloud_fn(rng)
    a = _actions(cloud, B, H, rep, rng)
    full = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, material))
    seq = full["state_seqs"]
    assert torch.isfinite(seq).all()
    # oracle on the first two candidates (the dense reference cannot run these sizes whole; the oracle can)
    want = O.dynamics(W, 3, cloud, a[:2], task)
    err = np.abs(seq[:2].cpu().numpy() - want["state_seqs"]).max()
    assert err <= POS_TOL, err
    # size-independent property: any sub-batch / chunking gives the same bits per candidate
    part = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a[1:4]).to(dev), m, dev, _ppm(task, material))
    assert torch.equal(part["state_seqs"], seq[1:4])
    m.engine(dev).set_chunk(2)
    again = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, material))
    m.engine(dev).set_chunk(0)
    assert torch.equal(again["state_seqs"], seq)


def test_repeat_zero_and_mixed_repeats(ag, O, dev):
    """action_repeat == 0 leaves zeros in state_seqs and the next look-ahead step starts from them (forward_dynamics.py:32,38)."""
    rng = np.random.default_rng(5)
    task = _task("rope")
    W, m = _model(ag, O, "rope", 5, dev)
    cloud = _rope(80, rng)
    a = _actions(cloud, 3, 2, [[0, 2], [3, 0], [1, 4]], rng)
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "rope"))
    want = O.dynamics(W, 3, cloud, a, task)
    got = out["state_seqs"].cpu().numpy()
    assert np.all(got[0, 0] == 0) and np.all(got[1, 1] == 0)
    assert np.abs(got - want["state_seqs"]).max() <= POS_TOL


def test_per_particle_physics_and_pstep1(ag, O, dev):
    rng = np.random.default_rng(6)
    task = _task("rope")
    cloud = _rope(90, rng)
    a = _actions(cloud, 2, 1, 3, rng)
    phys = rng.uniform(0.1, 0.9, 90).astype(np.float32)
    for pstep in (1, 3):
        W, m = _model(ag, O, "rope", 6 + pstep, dev, pstep)
        ppm = _ppm(task, "rope")
        ppm.physics_param = {"rope": torch.from_numpy(phys)}
        out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, ppm)
        want = O.dynamics(W, pstep, cloud, a, task, physics_param=phys)
        assert np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max() <= POS_TOL


def test_masked_rollout_with_holes_vs_oracle(ag, O, dev):
    """dynamics_masked with a NON-prefix mask: p_instance marks the first `count` rows (forward_dynamics.py:294-300),
    attrs follow the mask (:287) - both quirks are reproduced."""
    rng = np.random.default_rng(8)
    task = _task("rope", max_nR=4000)
    W, m = _model(ag, O, "rope", 8, dev)
    B, n = 3, 100
    state = np.zeros((B, n, 3), np.float32)
    mask = np.zeros((B, n), bool)
    for b in range(B):
        state[b] = _rope(n, rng)
        mask[b] = rng.uniform(size=n) > (0.0, 0.2, 0.35)[b]
        state[b, ~mask[b]] = 0
    a = _actions(state[0], B, 1, [3, 2, 4], rng)[:, 0]
    out = ag.dynamics_masked(torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a).to(dev),
                             m, dev, _ppm(task, "rope"))
    want = O.dynamics_masked(W, 3, state, mask, a, task)
    assert np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max() <= POS_TOL


def test_unsupported_and_invalid_inputs_raise(ag, O, dev):
    big = torch.zeros((1, 5000, 3), device=dev)
    ones = torch.ones((1, 5000), dtype=torch.bool, device=dev)
    with pytest.raises(NotImplementedError):
        ag.construct_edges_index(big, 0.5, ones, ~ones, 10, False, edge_cap=10)
    small = torch.rand((1, 400, 3), device=dev)
    ones = torch.ones((1, 400), dtype=torch.bool, device=dev)
    with pytest.raises(NotImplementedError):
        ag.construct_edges_index(small, 0.5, ones, ~ones, 200, False, edge_cap=10)
    mc, mat, ds = _cfg("rope")
    mc = dict(mc, nf_effect=128, nf_particle=128, nf_relation=128)
    with pytest.raises(NotImplementedError):
        ag.DynamicsPredictor(mc, mat, ds, dev)
    mc2, mat2, ds2 = _cfg("rope")
    with pytest.raises(NotImplementedError):
        ag.DynamicsPredictor(dict(mc2, offset_dim=3), mat2, ds2, dev)
    _, m = _model(ag, O, "rope", 1, dev)
    with pytest.raises(AssertionError):          # two physics keys (model.py:186-187)
        m(state=torch.zeros((1, 4, 3, 3), device=dev), attrs=torch.zeros((1, 3, 2), device=dev),
          Rr=torch.zeros((1, 1, 3), device=dev), Rs=torch.zeros((1, 1, 3), device=dev),
          p_instance=torch.ones((1, 2, 1), device=dev), action=torch.zeros((1, 3, 3), device=dev),
          a_physics_param=torch.zeros((1, 1)), b_physics_param=torch.zeros((1, 1)))
    task = _task("rope", pusher_points=[[0, 0, 0.1]] * 3, eef_num=3)
    with pytest.raises(NotImplementedError, match="pusher"):
        ag.dynamics(torch.zeros((10, 3), device=dev), torch.zeros((1, 1, 4), device=dev), m, dev, _ppm(task, "rope"))
    with pytest.raises(TypeError):
        ag.dynamics(torch.zeros((10, 3), device=dev), torch.zeros((1, 1, 4), device=dev), torch.nn.Linear(1, 1), dev,
                    _ppm(_task("rope"), "rope"))


def test_edge_overflow_reports_true_count(ag, dev):
    """ag_build_edges with a too-small edge_cap: nothing written, n_edges still the true count (pad_torch semantics)."""
    rng = np.random.default_rng(2)
    pos = torch.from_numpy(_grid(12, 0.1, 0.01, rng)[None]).to(dev)
    ones = torch.ones((1, 144), dtype=torch.bool, device=dev)
    big = ag.construct_edges_index(pos, 0.5, ones, ~ones, 10, False)
    small = ag.construct_edges_index(pos, 0.5, ones, ~ones, 10, False, edge_cap=100)
    assert int(small.n_edges[0]) == int(big.n_edges[0]) > 100
    assert torch.equal(small.row_ptr, big.row_ptr)


def test_empty_and_degenerate_graphs(ag, O, dev):
    """No valid particle at all / a single particle / everything out of radius."""
    pos = torch.rand((2, 6, 3), device=dev) * 100
    none = torch.zeros((2, 6), dtype=torch.bool, device=dev)
    el = ag.construct_edges_index(pos, 0.5, none, none, 3, False)
    assert el.n_edges.tolist() == [0, 0] and int(el.row_ptr.abs().sum()) == 0
    allv = ~none
    el = ag.construct_edges_index(pos, 1e-3, allv, none, 3, True)       # only self-loops survive
    for (r, s) in _edges_to_lists(el):
        assert np.array_equal(r, np.arange(6)) and np.array_equal(s, np.arange(6))
    Rr, Rs = el.to_dense()
    assert Rr.shape == (2, 6, 6)


# ------------------------------------------------------------------------------------------------- cost functions (§8(f) 1)
COST_TOL = 2e-5


def test_costs_vs_reference_golden(ag, dev):
    from functools import partial
    from helpers import load_golden
    g = load_golden("costs")
    t = lambda k: torch.from_numpy(np.asarray(g[k])).to(dev)
    B, H, N, _ = g["state"].shape
    flat = t("state").reshape(B * H, N, 3)
    assert np.abs(ag.chamfer(flat, t("target")[None]).cpu().numpy() - g["chamfer"]).max() < COST_TOL
    assert np.abs(ag.box_loss(flat, t("target_box")).cpu().numpy() - g["box_loss"]).max() < COST_TOL
    for kind in ("rope", "cloth", "granular"):
        got = getattr(ag, kind + "_penalty")(t("state"), t("action"), t("state_cur"), sim_real_ratio=10.0)
        assert np.abs(got.cpu().numpy() - g[kind + "_penalty"]).max() < COST_TOL, kind
    mc = ag.mean_chamfer(t("mc_pred"), t("mc_real"), t("mc_pred_mask"), t("mc_real_mask"))
    assert mc.dtype == np.float64 and np.abs(mc - g["mean_chamfer"]).max() < COST_TOL
    for err_name in ("chamfer", "box"):
        err = partial(ag.chamfer, y=t("target")[None]) if err_name == "chamfer" else partial(ag.box_loss, target=t("target_box"))
        for kind in ("rope", "cloth", "granular"):
            pen = partial(getattr(ag, kind + "_penalty"), sim_real_ratio=10.0)
            r = ag.running_cost(t("state"), t("action"), t("state_cur"), error_func=err, penalty_func=pen, bbox=g["bbox"])
            want = g[f"reward::{err_name}::{kind}"]
            assert np.abs(r["reward_seqs"].cpu().numpy() - want).max() < 5e-5 * max(1.0, np.abs(want).max()), (err_name, kind)
    # a caller's error / penalty callable may return CPU tensors (the reference's signature allows any callable): the values
    # are moved to the device before their addresses go to the kernel - same reward, no host pointer in a HIP launch
    err_cpu = lambda x: ag.chamfer(x, t("target")[None]).cpu()
    pen_cpu = lambda *a: ag.rope_penalty(*a, sim_real_ratio=10.0).cpu().double()
    r = ag.running_cost(t("state"), t("action"), t("state_cur"), error_func=err_cpu, penalty_func=pen_cpu, bbox=g["bbox"])
    assert np.abs(r["reward_seqs"].cpu().numpy() - g["reward::chamfer::rope"]).max() < 5e-5 * max(1.0, np.abs(g["reward::chamfer::rope"]).max())


def test_chamfer_full_size_vs_oracle(ag, dev):
    """BASELINE-size clouds (2025 predicted x 3000 target points): oracle on 3 rows; identity and symmetry properties."""
    from oracle import costs_oracle as Cc
    rng = np.random.default_rng(31)
    x = rng.normal(0, 1, (5, 2025, 3)).astype(np.float32)
    y = rng.normal(0.2, 1, (1, 3000, 3)).astype(np.float32)
    got = ag.chamfer(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
    want = Cc.chamfer(x[:3], y)
    assert np.abs(got[:3] - want).max() < COST_TOL
    same = ag.chamfer(torch.from_numpy(x).to(dev), torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.all(same == 0.0)                                         # chamfer(x, x) == 0 exactly
    xy = ag.chamfer(torch.from_numpy(x[:1]).to(dev), torch.from_numpy(y).to(dev))
    yx = ag.chamfer(torch.from_numpy(y).to(dev), torch.from_numpy(x[:1]).to(dev))
    assert abs(float(xy[0]) - float(yx[0])) < 1e-6                     # symmetric up to the order of the two means


# ------------------------------------------------------------------------------------------------- bf16x3 precision mode
@pytest.mark.parametrize("name,material", [("dyn_rope", "rope"), ("dyn_granular", "granular"), ("dyn_cloth", "cloth")])
def test_bf16x3_mode_meets_the_same_bar(ag, dev, name, material):
    """The opt-in bf16x3 arithmetic (3-way bf16 split, fp32 accumulate) against the reference's golden rollouts at the
    SAME 1e-5 tolerance, and against the exact-fp32 mode (difference far below the tolerance)."""
    from helpers import load_golden, task_of
    from test_gpu_parity import _model as golden_model
    g = load_golden(name)
    task = task_of(g)
    m = golden_model(ag, g, material, dev)
    s0, a = torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev)
    m.set_precision("fp32")                                            # whatever AG_PRECISION says
    exact = ag.dynamics(s0, a, m, dev, _ppm(task, material))["state_seqs"]
    m.set_precision("bf16x3")
    fast = ag.dynamics(s0, a, m, dev, _ppm(task, material))["state_seqs"]
    m.set_precision("fp32")
    again = ag.dynamics(s0, a, m, dev, _ppm(task, material))["state_seqs"]
    assert torch.equal(again, exact)                                   # switching back restores the exact path bit for bit
    err_ref = np.abs(fast.cpu().numpy() - g["state_seqs"]).max()
    err_exact = float((fast - exact).abs().max())
    print(f"{name}: bf16x3 vs reference {err_ref:.2e}, vs exact-fp32 mode {err_exact:.2e}")
    assert err_ref <= POS_TOL and err_exact <= 2e-6


def test_bf16x3_twenty_step_free_running(ag, O, dev):
    rng = np.random.default_rng(21)
    task = _task("granular", max_nR=20000)
    W, m = _model(ag, O, "granular", 21, dev)
    cloud = _grid(18, 0.12, 0.02, rng)
    a = _actions(cloud, 2, 2, 10, rng, spread=0.5)
    want = O.dynamics(W, 3, cloud, a, task)
    m.set_precision("bf16x3")
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "granular"))
    m.set_precision("fp32")
    err = np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max()
    print(f"bf16x3 20-step free-running error vs oracle {err:.2e}")
    assert err <= POS_TOL


def test_ppm_dynamics_error_vs_reference_golden(ag, dev):
    """the physics-parameter optimiser's objective (physics_param_optimizer.py:178-226)."""
    from helpers import load_golden, task_of
    from test_gpu_parity import _model as golden_model
    g = load_golden("ppm_dynamics_error")
    task = task_of(g)
    m = golden_model(ag, g, "rope", dev)
    ppm = _ppm(task, "rope")
    ppm.model, ppm.device = m, dev
    n = int(g["n_act"])
    inits, reals, acts = ([g[f"{k}{i}"] for i in range(n)] for k in ("init", "real", "act"))
    for v, want in zip(g["phys_values"], g["errors"]):
        got = float(ag.dynamics_error([float(v)], ppm, inits, reals, acts))
        assert abs(got - want) < 2e-5, (v, got, want)


def test_mpc_iteration_end_to_end(ag, O, dev):
    """one whole MPPI iteration on the engine (sample -> rollout -> running_cost -> update -> best)."""
    from functools import partial
    rng = np.random.default_rng(12)
    task = _task("rope")
    W, m = _model(ag, O, "rope", 12, dev)
    cloud = _rope(120, rng)
    s0 = torch.from_numpy(cloud).to(dev)
    ppm = _ppm(task, "rope")
    lo = torch.tensor([cloud[:, 0].min() - 0.3, cloud[:, 2].min() - 0.3, -3.14, 2.0], device=dev)
    hi = torch.tensor([cloud[:, 0].max() + 0.3, cloud[:, 2].max() + 0.3, 3.14, 4.0], device=dev)
    target = torch.from_numpy(cloud + np.float32([0.2, 0, 0.1])).to(dev)
    rollout = partial(ag.dynamics, model=m, device=dev, ppm_optimizer=ppm)
    evaluate = partial(ag.running_cost, error_func=partial(ag.chamfer, y=target[None]),
                       penalty_func=partial(ag.rope_penalty, sim_real_ratio=10.0),
                       bbox=np.array([[-4.5, 0.0], [-2.5, 4.5]]))
    torch.manual_seed(0)
    act_seq = torch.rand((2, 4), device=dev) * (hi - lo) + lo
    out = ag.mpc_iteration(s0, act_seq, rollout, evaluate, lo, hi, n_sample=48, device=dev, reward_weight=500.0)
    r = out["reward_seqs"]
    assert r.shape == (48,) and torch.isfinite(r).all()
    assert float(out["best_reward"]) == float(r.max())
    # the re-rolled best candidate reproduces its reward (same candidate, batch of one: bit-identical rollout)
    assert out["best_model_output"]["state_seqs"].shape == (1, 2, 120, 3)
    lo_c, hi_c = lo.cpu(), hi.cpu()
    a = out["mppi_act_seq"].cpu()
    assert a.shape == (2, 4) and bool(((a >= lo_c - 1e-6) & (a <= hi_c + 1e-6)).all())
    # reference check of the whole chain on the best candidate with the oracle
    best = out["act_seq"].cpu().numpy()[None]
    want = O.dynamics(W, 3, cloud, best, task)["state_seqs"]
    assert np.abs(out["best_model_output"]["state_seqs"].cpu().numpy() - want).max() <= POS_TOL
    # several update iterations (planner.py:240-260): iteration i perturbs the nominal sequence of iteration i-1, the best
    # candidate over ALL iterations is returned - so its reward cannot be below the single-iteration result's
    torch.manual_seed(5)
    one = ag.mpc_iteration(s0, act_seq, rollout, evaluate, lo, hi, n_sample=48, device=dev, reward_weight=500.0,
                           noise_level=0.3, rollout_best=False)
    torch.manual_seed(5)                                                # same first-iteration samples
    more = ag.mpc_iteration(s0, act_seq, rollout, evaluate, lo, hi, n_sample=48, device=dev, reward_weight=500.0,
                            noise_level=0.3, n_update_iter=3)
    assert float(more["best_reward"]) >= float(one["best_reward"])
    again = evaluate(rollout(s0, more["act_seq"][None])["state_seqs"], more["act_seq"][None], state_cur=s0)["reward_seqs"]
    assert more["best_model_output"]["state_seqs"].shape == (1, 2, 120, 3) and torch.isfinite(again).all()


def test_single_graph_builder_vs_reference_golden(ag, dev):
    """construct_edges_from_states (default path), incl. the double-precision threshold quirk."""
    import json
    from helpers import load_golden, split_edges
    g = load_golden("edges_single")
    cases = json.loads(bytes(g["cases_json"]).decode())
    for ci, c in enumerate(cases):
        pre = f"case{ci}::"
        args = (torch.from_numpy(g[pre + "states"]).to(dev), c["adj_thresh"], torch.from_numpy(g[pre + "mask"]).to(dev),
                torch.from_numpy(g[pre + "tool_mask"]).to(dev))
        el = ag.construct_edges_from_states(*args, topk=c["topk"], connect_tools_all=c["connect_tools_all"], as_index=True)
        (r, s), = _edges_to_lists(el)
        (wr, ws), = split_edges(g, pre)
        assert np.array_equal(r, wr) and np.array_equal(s, ws), ci
        Rr, Rs = ag.construct_edges_from_states(*args, topk=c["topk"], connect_tools_all=c["connect_tools_all"])
        assert Rr.shape == (len(wr), g[pre + "states"].shape[0]) and np.array_equal(Rr.argmax(-1).cpu().numpy(), wr)
    # last case: a pair at distance^2 == fp32(0.16) exactly, adj_thresh 0.4.  The single-graph builder (double-precision
    # square) leaves it unconnected; the batch builder (fp32 square = 0.16000001) connects it (SURVEY a5'(ii)).
    assert len(wr) == 3
    elb = ag.construct_edges_index(args[0][None], 0.4, args[2][None], args[3][None], 3, False)
    assert int(elb.n_edges[0]) == 5


@pytest.mark.parametrize("material,n_max,cloud_fn", [
    ("rope", 150, lambda n, r: _rope(n, r)),
    ("granular", 160, lambda n, r: _grid(13, 0.12, 0.02, r)[:n]),
    ("cloth", 169, lambda n, r: _grid(13, 0.3, 0.02, r)[:n]),
])
def test_mixed_variable_size_graphs_masked(ag, O, dev, material, n_max, cloud_fn):
    """BASELINE configs[4] in miniature: variable-size graphs (prefix masks of different lengths) of every material
    through dynamics_masked - 5-point pusher (granular) and connect_tools_all + gripper (cloth) in masked mode."""
    rng = np.random.default_rng(50 + n_max)
    task = _task(material, max_nR=20000)
    W, m = _model(ag, O, material, 50 + n_max, dev)
    counts = [n_max, n_max // 2, (3 * n_max) // 4, n_max - 1]
    B = len(counts)
    state = np.zeros((B, n_max, 3), np.float32)
    mask = np.zeros((B, n_max), bool)
    for b, c in enumerate(counts):
        state[b, :c] = cloud_fn(c, rng)
        mask[b, :c] = True
    a = _actions(state[0], B, 1, [2, 3, 1, 4], rng, spread=0.4)[:, 0]
    out = ag.dynamics_masked(torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a).to(dev),
                             m, dev, _ppm(task, material))
    want = O.dynamics_masked(W, 3, state, mask, a, task)
    err = np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max()
    assert err <= POS_TOL, err


def _rule_kwargs(c):
    kw = {k: np.float32(v) for k, v in c["bounds_f32"].items()}        # numpy float32 scalars, as the eval rollout passes
    kw.update(topk=c["topk"], connect_tools_all=c["connect_tools_all"], connect_tools_surface=c["connect_tools_surface"],
              connect_tool_all_non_fixed=c["connect_tool_all_non_fixed"], kNN=c["kNN"])
    return kw


def test_single_graph_tool_rules_vs_reference_golden(ag, dev):
    """SURVEY 8(f) rank 3, second half: the tool rules of construct_edges_from_states (graph.py:125-221) - non-fixed
    particles, flat kNN filter, two closest surface planes - bit-exact against the reference's own output."""
    import json
    from helpers import load_golden, split_edges
    g = load_golden("edges_single_rules")
    meta = json.loads(bytes(g["meta_json"]).decode())
    for ci, c in enumerate(meta["cases"]):
        pre = f"case{ci}::"
        args = (torch.from_numpy(g[pre + "states"]).to(dev), c["adj_thresh"], torch.from_numpy(g[pre + "mask"]).to(dev),
                torch.from_numpy(g[pre + "tool_mask"]).to(dev))
        el = ag.construct_edges_from_states(*args, as_index=True, **_rule_kwargs(c))
        (r, s), = _edges_to_lists(el)
        (wr, ws), = split_edges(g, pre)
        assert np.array_equal(r, wr) and np.array_equal(s, ws), ci
        rp = el.row_ptr[0].cpu().numpy()
        assert rp[0] == 0 and rp[-1] == len(wr) and np.array_equal(np.diff(rp), np.bincount(wr, minlength=el.N))
        Rr, Rs = ag.construct_edges_from_states(*args, **_rule_kwargs(c))
        assert Rr.shape == (c["n_rel"], g[pre + "states"].shape[0])
        assert np.array_equal(Rs.argmax(-1).cpu().numpy(), ws)


def test_single_graph_tool_rules_vs_oracle_sweep(ag, O, dev):
    """Random blobs over the rule switches (incl. tools in the middle of the index range and invalid particles)."""
    rng = np.random.default_rng(5)
    for trial in range(12):
        N_o, M = int(rng.integers(40, 400)), int(rng.integers(1, 6))
        N = N_o + M
        pos = rng.uniform(0, 1, (N, 3)).astype(np.float32) * np.float32([1.0, 0.3, 1.0])
        tool = np.zeros(N, bool)
        tool[rng.choice(N, M, replace=False)] = True                    # tools anywhere in the index range
        mask = rng.uniform(size=N) > 0.05
        mask[tool] = True
        obj = pos[mask & ~tool]
        kw = dict(topk=int(rng.integers(3, 12)), connect_tools_all=bool(trial & 1), kNN=float(rng.choice([1.0, 0.7, 0.35, 0.05])),
                  connect_tool_all_non_fixed=bool(trial % 3), connect_tools_surface=bool(trial % 4 < 2),
                  max_y=np.max(obj[:, 1]) * 0.8, min_y=np.min(obj[:, 1]), max_x=np.max(obj[:, 0]) * 0.8,
                  min_x=np.min(obj[:, 0]) + 0.2, max_z=np.max(obj[:, 2]) * 0.8, min_z=np.min(obj[:, 2]) + 0.2)
        thr = float(rng.choice([0.12, 0.2, 0.3]))
        wr, ws = O.construct_edges_from_states(pos, thr, mask, tool, check_ties=True, **kw)
        el = ag.construct_edges_from_states(torch.from_numpy(pos).to(dev), thr, torch.from_numpy(mask).to(dev),
                                            torch.from_numpy(tool).to(dev), as_index=True, **kw)
        (r, s), = _edges_to_lists(el)
        assert np.array_equal(r, wr) and np.array_equal(s, ws), (trial, kw)


def test_backoff_loop_vs_reference_golden(ag, dev):
    """rollout.py:185-222: kNN shrinks by knn_increment to min_kNN, then top-k drops, until the graph fits max_nR."""
    import json
    from helpers import load_golden, split_edges
    g = load_golden("edges_single_rules")
    b = json.loads(bytes(g["meta_json"]).decode())["backoff"]
    args = (torch.from_numpy(g["backoff::states"]).to(dev), b["adj_thresh"], torch.from_numpy(g["backoff::mask"]).to(dev),
            torch.from_numpy(g["backoff::tool_mask"]).to(dev))
    Rr, Rs = ag.construct_edges_with_backoff(*args, topk=b["topk"], max_nR=b["max_nR"], knn_thresh=b["knn_thresh"],
                                             min_kNN=b["min_kNN"], knn_increment=b["knn_increment"],
                                             max_y=np.float32(b["max_y"]), min_y=np.float32(b["min_y"]))
    (wr, ws), = split_edges(g, "backoff::")
    assert Rr.shape == (b["max_nR"], g["backoff::states"].shape[0])
    n = len(wr)
    assert n == b["trail"][-1][2]
    assert np.array_equal(Rr[:n].argmax(-1).cpu().numpy(), wr) and np.array_equal(Rs[:n].argmax(-1).cpu().numpy(), ws)
    assert float(Rr[n:].abs().sum()) == 0.0 and float(Rs[n:].abs().sum()) == 0.0
    with pytest.raises(Exception, match="Exceeds max dims"):            # nothing left to shrink: the reference's loop dies the same way
        ag.construct_edges_with_backoff(*args, topk=1, max_nR=3, knn_thresh=0.1, min_kNN=0.2,
                                        max_y=np.float32(b["max_y"]), min_y=np.float32(b["min_y"]))


def test_rollout_radius_only_graph(ag, O, dev):
    """topk >= N: no top-k, the rollout takes the CSR (emit) path instead of the slot-indexed sender lists."""
    rng = np.random.default_rng(21)
    task = _task("rope", topk=500, adj_thresh=0.12, max_nR=20000)
    W, m = _model(ag, O, "rope", 21, dev)
    cloud = _rope(120, rng)
    a = _actions(cloud, 3, 2, [2, 1, 3], rng)
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "rope"))
    want = O.dynamics(W, 3, cloud, a, task)
    assert np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max() <= POS_TOL


def test_rollout_at_the_particle_limit(ag, O, dev):
    """4095 object particles + 1 tool = the edge builder's 4096-particle limit, one rollout step, against the oracle;
    one particle more is refused loudly."""
    rng = np.random.default_rng(22)
    task = _task("cloth", max_nR=40000)
    W, m = _model(ag, O, "cloth", 22, dev)
    cloud = _grid(64, 0.3, 0.02, rng)[:4095]
    a = _actions(cloud, 2, 1, 1, rng, spread=2.0)
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "cloth"))
    want = O.dynamics(W, 3, cloud, a, task)
    assert np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max() <= POS_TOL
    too_many = _grid(65, 0.3, 0.02, rng)[:4096]
    with pytest.raises(NotImplementedError):
        ag.dynamics(torch.from_numpy(too_many).to(dev), torch.from_numpy(a).to(dev), m, dev, _ppm(task, "cloth"))


def test_repeated_calls_are_bit_identical_across_workspace_recarves(ag, O, dev):
    """Two in-library streams, a slab workspace that is re-carved whenever the batch shape changes: the same inputs must
    give the same bits on every call (no float atomics, no stale scratch)."""
    rng = np.random.default_rng(23)
    task = _task("cloth")
    W, m = _model(ag, O, "cloth", 23, dev)
    cloud = _grid(40, 0.3, 0.02, rng)                                    # 1600 particles: 96 x 1601 rows use both streams
    a = torch.from_numpy(_actions(cloud, 96, 2, 2, rng, spread=2.0)).to(dev)
    s0 = torch.from_numpy(cloud).to(dev)
    ref = None
    for it in range(6):
        out = ag.dynamics(s0, a, m, dev, _ppm(task, "cloth"))["state_seqs"]
        assert torch.isfinite(out).all()
        ref = out.clone() if ref is None else ref
        assert torch.equal(out, ref), it
        ag.dynamics(s0, a[: 5 + 7 * it], m, dev, _ppm(task, "cloth"))   # another shape in between


def test_forward_on_overflowed_edgelist_raises_instead_of_faulting(ag, O, dev):
    """An EdgeList built with a too-small edge_cap carries the TRUE edge count and unwritten index arrays
    (pad_torch semantics); forward() must refuse it like the reference's pad_torch does (utils.py:63-65)."""
    rng = np.random.default_rng(2)
    cloud = _grid(12, 0.1, 0.01, rng)
    N = cloud.shape[0]
    _, m = _model(ag, O, "rope", 2, dev)
    st = torch.from_numpy(np.repeat(cloud[None, None], 4, 1)).to(dev)   # (1, 4, N, 3)
    ones = torch.ones((1, N), dtype=torch.bool, device=dev)
    args = dict(state=st, attrs=torch.cat([torch.ones(1, N, 1), torch.zeros(1, N, 1)], -1).to(dev),
                p_instance=torch.ones((1, N - 1, 1), device=dev), action=torch.zeros((1, N, 3), device=dev),
                rope_physics_param=torch.full((1, 1), 0.5, device=dev))
    small = ag.construct_edges_index(st[:, -1], 0.5, ones, ~ones, 10, False, edge_cap=100)
    small.recv.fill_(2 ** 30)                                           # what torch.empty may hold: wild indices
    small.send.fill_(2 ** 30)
    with pytest.raises(Exception, match="Exceeds max dims"):
        m(edges=small, **args)
    ok = ag.construct_edges_index(st[:, -1], 0.5, ones, ~ones, 10, False)
    pos, _ = m(edges=ok, **args)                                        # the ctx is still usable
    assert torch.isfinite(pos).all()


_FAIL_CHILD = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import adaptigraph_amd as ag
from adaptigraph_amd import _lib
from oracle import adaptigraph_oracle as O
from test_gpu_parity import _ppm
from test_gpu_more import _task, _grid, _actions, _model
assert _lib.LIB_PATH.endswith("_diag.so")
dev = torch.device("cuda:0")
rng = np.random.default_rng(29)
task = _task("cloth")
cloud = _grid(40, 0.3, 0.02, rng)                                       # 96 x 1601 rows: two streams
a = torch.from_numpy(_actions(cloud, 96, 1, 2, rng, spread=2.0)).to(dev)
s0 = torch.from_numpy(cloud).to(dev)
os.environ.pop("AG_TEST_FAIL_AT_CHUNK", None)
_, m = _model(ag, O, "cloth", 29, dev)                                  # context created WITHOUT the hook
good = ag.dynamics(s0, a, m, dev, _ppm(task, "cloth"))["state_seqs"].clone()
os.environ["AG_TEST_FAIL_AT_CHUNK"] = "1"                              # the hook is read when a context is created
_, bad = _model(ag, O, "cloth", 29, dev)
bad.engine(dev)                                                         # (the context is created on first use)
del os.environ["AG_TEST_FAIL_AT_CHUNK"]
try:
    ag.dynamics(s0, a, bad, dev, _ppm(task, "cloth"))
    sys.exit("no failure was injected")
except RuntimeError as e:
    assert "injected failure" in str(e), e
torch.cuda.synchronize()                                                # nothing of the failed call is left in flight
# the failed context itself must be usable: it is driven again with a batch of one chunk (chunk 1 is never reached)
one = ag.dynamics(s0, a[:16], bad, dev, _ppm(task, "cloth"))["state_seqs"]   # (16 x 1601 rows: below the two-stream threshold)
assert torch.equal(one, good[:16])
again = ag.dynamics(s0, a, m, dev, _ppm(task, "cloth"))["state_seqs"]  # and the other context of the process is unaffected
assert torch.equal(again, good)
print("CHILD_OK")
'''


def test_failed_rollout_joins_its_streams_and_leaves_the_ctx_usable(dev):
    """A failure in the middle of the chunk loop (after the fork onto the second stream) must join the streams back; the
    failed context stays usable and a second context of the process is untouched.  The failure is injected by the
    DIAGNOSTIC build of the library (-DAG_DIAG, AG_TEST_FAIL_AT_CHUNK; the product library has no such hook), loaded in a
    child process through ADAPTIGRAPH_AMD_LIB."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "adaptigraph_amd", "csrc", "libadaptigraph_hip_diag.so")
    assert os.path.exists(diag), "build the diagnostic library: adaptigraph_amd/csrc/build.sh diag"
    env = dict(os.environ, ADAPTIGRAPH_AMD_LIB=diag)
    r = subprocess.run([sys.executable, "-c", _FAIL_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("topk", [10, 500])
def test_ragged_row_list_equals_dense_rows_bitwise(ag, O, dev, topk):
    """Masked batches run the propagate chains over a compact row list + one phantom candidate for the masked-out
    particles; option ragged=0 runs all B x N rows as before.  Same bits for EVERY row, masked-out ones included (the
    reference moves those too, model.py:338), with masks that have holes, an all-valid and a nearly empty candidate;
    topk 500 takes the CSR (radius-only) graph path."""
    import os
    rng = np.random.default_rng(61)
    task = _task("rope", max_nR=20000, topk=topk, adj_thresh=0.5 if topk == 10 else 0.12)
    W, m = _model(ag, O, "rope", 61, dev)
    B, n = 7, 150
    state = np.zeros((B, n, 3), np.float32)
    mask = np.zeros((B, n), bool)
    keep = [1.0, 0.8, 0.5, 0.97, 0.3, 1.0, 0.02]
    for b in range(B):
        state[b] = _rope(n, rng)
        mask[b] = rng.uniform(size=n) < keep[b]
        mask[b, 0] = True
        state[b, ~mask[b]] = rng.normal(0, 1, (int((~mask[b]).sum()), 3))      # padding rows need not be zero
    a = _actions(state[0], B, 1, [3, 2, 4, 1, 5, 2, 3], rng)[:, 0]
    args = (torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a).to(dev))
    ragged = ag.dynamics_masked(*args, m, dev, _ppm(task, "rope"))["state_seqs"]
    assert m.engine(dev).rollout_counts() == (20, 20)                    # repeats 3+2+4+1+5+2+3: the live prefix shrinks per step
    with m.engine(dev).options(repeat_sort=0):
        unsorted = ag.dynamics_masked(*args, m, dev, _ppm(task, "rope"))["state_seqs"]
        assert m.engine(dev).rollout_counts() == (35, 20)                # every slot stepped to the chunk maximum (5)
    assert torch.equal(unsorted, ragged)
    with m.engine(dev).options(ragged=0):
        dense = ag.dynamics_masked(*args, m, dev, _ppm(task, "rope"))["state_seqs"]
    assert torch.isfinite(ragged).all() and torch.equal(ragged, dense)
    m.engine(dev).set_chunk(3)                                          # phantom slot moves with the chunk size
    try:
        chunked = ag.dynamics_masked(*args, m, dev, _ppm(task, "rope"))["state_seqs"]
    finally:
        m.engine(dev).set_chunk(0)
    assert torch.equal(chunked, dense)
    want = O.dynamics_masked(W, 3, state, mask, a, task)["state_seqs"]
    assert np.abs(ragged.cpu().numpy() - want).max() <= POS_TOL


@pytest.mark.parametrize("material,cloud_fn,B", [
    ("rope", lambda r: _rope(120, r), 3),
    ("granular", lambda r: _grid(12, 0.12, 0.02, r), 2),
    ("cloth", lambda r: _grid(14, 0.3, 0.02, r), 2),
])
def test_latency_kernels_equal_throughput_kernels_bitwise(ag, O, dev, material, cloud_fn, B):
    """Small batches run the latency-mode chains (csrc/ag_lat.hip: 32-row workgroups, every layer split over four
    wavefronts on v_mfma_f32_16x16x4_f32), large ones the throughput chains (32-row wavefronts, v_mfma_f32_32x32x2_f32).
    Both feed the k's of every layer in the same order, so they must agree bit for bit - which is what keeps results
    independent of batch size, chunking and sharding across the switch."""
    import os
    rng = np.random.default_rng(71)
    task = _task(material)
    W, m = _model(ag, O, material, 71, dev)
    cloud = cloud_fn(rng)
    a = torch.from_numpy(_actions(cloud, B, 2, [[2, 1], [3, 2], [1, 3]][:B], rng)).to(dev)
    s0 = torch.from_numpy(cloud).to(dev)
    out = {}
    for mode in ("0", "1"):
        with m.engine(dev).options(latency=int(mode)):
            out[mode] = ag.dynamics(s0, a, m, dev, _ppm(task, material))["state_seqs"]
    assert torch.isfinite(out["1"]).all()
    assert torch.equal(out["0"], out["1"])
    auto = ag.dynamics(s0, a, m, dev, _ppm(task, material))["state_seqs"]      # by size: latency mode here
    assert torch.equal(auto, out["0"])
    want = O.dynamics(W, 3, cloud, a.cpu().numpy(), task)["state_seqs"]
    assert np.abs(auto.cpu().numpy() - want).max() <= POS_TOL


@pytest.mark.parametrize("N_o,M,topk,thr,pitch,cta", [
    (2025, 1, 5, 0.75, 0.3, True),        # the benchmarked cloth shape: two 1013-row slices per candidate
    (1024, 5, 20, 0.40, 0.12, True),      # granular: k = 20 kept in registers per lane
    (300, 1, 10, 0.5, 0.05, False),       # rope-sized: five blocks, most lanes of the last one idle
    (640, 2, 3, 0.3, 0.1, True),          # k below the smallest register tile (5)
])
def test_edge_block_schedule_equals_row_schedule_bitwise(ag, dev, N_o, M, topk, thr, pitch, cta):
    """Large batches build the top-k graph 64 receiver rows per wavefront (one row per lane, hit masks + per-lane sorted
    top-k: csrc/ag_edges.hip block_topk); option edge_block_min switches back to one row per wavefront.  Same edge lists,
    same CSR, same degrees - with holes in the masks, an empty candidate tail, far-away tools and exact distance ties."""
    import os
    rng = np.random.default_rng(N_o + topk)
    B, N = 130, N_o + M
    side = int(np.ceil(np.sqrt(N_o)))
    g = np.arange(side) * pitch
    xx, zz = np.meshgrid(g, g, indexing="ij")
    base = np.stack([xx.ravel(), np.zeros(side * side), zz.ravel()], 1)[:N_o].astype(np.float32)
    states = np.zeros((B, N, 3), np.float32)
    mask = np.ones((B, N), bool)
    tool = np.zeros((B, N), bool)
    tool[:, N_o:] = True
    for b in range(B):
        jitter = 0.0 if b % 9 == 0 else pitch / 6               # every ninth candidate is an exact lattice: mass ties
        states[b, :N_o] = base + rng.normal(0, 1, base.shape).astype(np.float32) * np.float32(jitter)
        states[b, N_o:] = states[b, rng.integers(0, N_o, M)] + rng.normal(0, pitch / 3, (M, 3))
        if b % 4 == 1:
            mask[b, :N_o] = rng.uniform(size=N_o) < 0.7          # holes
        if b % 4 == 2:
            mask[b, N_o // 3:N_o] = False                        # ragged tail
        if b % 5 == 3:
            states[b, N_o:, 0] += 1e3                            # tool out of reach: connect_tools_all flag stays off
    args = (torch.from_numpy(states).to(dev), thr, torch.from_numpy(mask).to(dev), torch.from_numpy(tool).to(dev), topk, cta)
    out = {}
    for mode, val in (("blocks", 1), ("rows", 1000000000)):
        with ag.default_engine(dev).options(edge_block_min=val):
            el = ag.construct_edges_index(*args)
            out[mode] = [t.cpu().numpy().copy() for t in (el.n_edges, el.recv, el.send, el.row_ptr)]
    n = out["rows"][0]
    assert n.min() > 0 and np.array_equal(out["blocks"][0], n)
    assert np.array_equal(out["blocks"][3], out["rows"][3])
    for b in range(B):
        assert np.array_equal(out["blocks"][1][b, :n[b]], out["rows"][1][b, :n[b]]), b
        assert np.array_equal(out["blocks"][2][b, :n[b]], out["rows"][2][b, :n[b]]), b


def test_planner_class_chunk_loop_equals_chunked_entry_on_the_engine(ag, O, dev):
    """The reference's planning loop (plan.py:177-247): a Planner configured with dynamics / running_cost / the MPPI
    helpers, called once per chunk of n_sample_chunk candidates, merged with merge_res.  Planner.trajectory_optimization_chunked
    does the same with one rollout call for all chunks and one for the winners - bit-identical result (a candidate's rollout
    does not depend on its batch), identical generator state afterwards."""
    from functools import partial
    from adaptigraph_amd.planner import Planner
    rng = np.random.default_rng(21)
    task = _task("rope")
    W, m = _model(ag, O, "rope", 21, dev)
    cloud = _rope(100, rng)
    s0 = torch.from_numpy(cloud).to(dev)
    ppm = _ppm(task, "rope")
    lo = torch.tensor([cloud[:, 0].min() - 0.3, cloud[:, 2].min() - 0.3, -3.14, 2.0], device=dev)
    hi = torch.tensor([cloud[:, 0].max() + 0.3, cloud[:, 2].max() + 0.3, 3.14, 4.0], device=dev)
    target = torch.from_numpy(cloud + np.float32([0.2, 0, 0.1])).to(dev)
    S, n_chunk, H = 24, 4, 2
    calls = []

    def rollout(state_cur, act_seqs):
        calls.append(int(act_seqs.shape[0]))
        return ag.dynamics(state_cur, act_seqs, model=m, device=dev, ppm_optimizer=ppm)

    cfg = {"action_dim": 4, "model_rollout_fn": rollout,
           "evaluate_traj_fn": partial(ag.running_cost, error_func=partial(ag.chamfer, y=target[None]),
                                       penalty_func=partial(ag.rope_penalty, sim_real_ratio=10.0),
                                       bbox=np.array([[-4.5, 0.0], [-2.5, 4.5]])),
           "sampling_action_seq_fn": partial(ag.sample_action_seq, action_lower_lim=lo, action_upper_lim=hi, n_sample=S,
                                             device=dev, noise_level=0.3, push_length=task["push_length"]),
           "clip_action_seq_fn": partial(ag.clip_actions, action_lower_lim=lo, action_upper_lim=hi),
           "optimize_action_mppi_fn": partial(ag.optimize_action_mppi, reward_weight=500.0, action_lower_lim=lo,
                                              action_upper_lim=hi, push_length=task["push_length"]),
           "n_sample": S, "n_look_ahead": H, "n_update_iter": 1, "reward_weight": 500.0, "action_lower_lim": lo,
           "action_upper_lim": hi, "planner_type": "MPPI", "device": dev, "verbose": False, "noise_level": 0.3,
           "rollout_best": True}
    planner = Planner(cfg)
    torch.manual_seed(3)
    act_seq = torch.rand((H, 4), device=dev) * (hi - lo) + lo
    torch.manual_seed(4)
    res_all = []
    planner.total_chunks = n_chunk                                      # plan.py:210
    for ci in range(n_chunk):                                           # plan.py:241-247
        planner.chunk_id = ci
        res = planner.trajectory_optimization(s0, act_seq)
        res_all.append({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in res.items()})
    loop = planner.merge_res(res_all)
    gen_after_loop = torch.cuda.get_rng_state(dev)
    assert calls == [S, 1] * n_chunk
    calls.clear()
    torch.manual_seed(4)
    fused = planner.trajectory_optimization_chunked(s0, act_seq, n_chunk)
    assert calls == [S * n_chunk, n_chunk]
    assert torch.equal(torch.cuda.get_rng_state(dev), gen_after_loop)
    assert torch.equal(fused["act_seq"], loop["act_seq"])
    assert torch.equal(fused["best_model_output"]["state_seqs"], loop["best_model_output"]["state_seqs"])
    assert torch.equal(fused["best_eval_output"]["reward_seqs"], loop["best_eval_output"]["reward_seqs"])
    assert fused["best_model_output"]["state_seqs"].shape == (1, H, 100, 3)
    # the winner against the oracle
    want = O.dynamics(W, 3, cloud, loop["act_seq"].cpu().numpy()[None], task)["state_seqs"]
    assert np.abs(loop["best_model_output"]["state_seqs"].cpu().numpy() - want).max() <= POS_TOL


def test_edge_chain_persistent_workgroups_equal_one_workgroup_per_tile_bitwise(ag, O, dev):
    """Option enc_persist=n runs k_edge_enc with n persistent workgroups walking the tiles (a measured-and-documented knob,
    DESIGN.md section 3.1); rows are independent columns of the MFMA, so the result must not change - also when n does not
    divide the tile count and some workgroups get one tile more."""
    import os
    rng = np.random.default_rng(81)
    task = _task("cloth")
    W, m = _model(ag, O, "cloth", 81, dev)
    cloud = _grid(20, 0.3, 0.02, rng)
    a = torch.from_numpy(_actions(cloud, 12, 2, [[2, 1]] * 12, rng)).to(dev)
    s0 = torch.from_numpy(cloud).to(dev)
    eng = m.engine(dev)
    with eng.options(latency=0):                                         # throughput chains at this size
        ref = ag.dynamics(s0, a, m, dev, _ppm(task, "cloth"))["state_seqs"]
        for n in (7, 64):
            with eng.options(enc_persist=n):
                got = ag.dynamics(s0, a, m, dev, _ppm(task, "cloth"))["state_seqs"]
            assert torch.isfinite(got).all() and torch.equal(got, ref), n


# ------------------------------------------------------------------------------------------------- repeat-aware launch order
@pytest.mark.parametrize("material,cloud_fn,B,H", [
    ("rope", lambda r: _rope(200, r), 500, 1),                          # the shipped planner's chunk: 500 x (200+1), H = 1
    ("cloth", lambda r: _grid(30, 0.3, 0.02, r), 150, 2),               # 150 x 901 rows: two streams, H = 2
])
def test_repeat_sorted_launch_order_is_bit_identical_and_runs_no_surplus_forward(ag, O, dev, material, cloud_fn, B, H):
    """ag_rollout orders the candidates of a launch chunk by action_repeat and launches every step over the prefix that is
    still live; option repeat_sort=0 steps every candidate of a chunk to the chunk maximum (the reference steps the whole
    batch to the batch maximum and discards the surplus, forward_dynamics.py:156-161).  Candidates are independent, so the
    outputs must be identical bit for bit - mixed repeats 2..15 (incl. a repeat of 0), one / two streams, several chunks -
    and the executed candidate-forwards must equal sum(action_repeat)."""
    rng = np.random.default_rng(97)
    task = _task(material, max_nR=40000)
    W, m = _model(ag, O, material, 97, dev)
    cloud = cloud_fn(rng)
    reps = rng.integers(2, 16, (B, H))
    if H == 1:
        reps[3, 0] = 0                                                  # never live: its slot stays zero (:32,:160)
    a_np = _actions(cloud, B, H, reps, rng, spread=0.8)
    if H == 1:
        a_np[3, 0, 3] = 0.5
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, material)
    eng = m.engine(dev)
    need = int(reps.sum())
    outs = {}
    for streams in (1, 2):
        for chunk in (0, 37):
            with eng.options(streams=streams, share_prefix=0):       # (this test counts the forwards of the repeat-aware order alone)
                eng.set_chunk(chunk)
                try:
                    got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
                    ex, nd = eng.rollout_counts()
                    assert nd == need and ex == need, (streams, chunk, ex, nd, need)
                    with eng.options(repeat_sort=0):
                        ref = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
                        ex0, nd0 = eng.rollout_counts()
                finally:
                    eng.set_chunk(0)
            assert nd0 == need and ex0 > need                            # the surplus the sorted order does not run
            assert torch.isfinite(got).all() and torch.equal(got, ref), (streams, chunk)
            outs[(streams, chunk)] = got
    first = outs[(1, 0)]
    assert all(torch.equal(first, o) for o in outs.values())
    if H == 1:
        assert float(first[3, 0].abs().max()) == 0.0
    print(f"{material}: sum(action_repeat) = {need} candidate-forwards; unsorted order executed {ex0} ({ex0 / need:.2f}x)")
    picks = [0, 3, B - 1]
    want = O.dynamics(W, 3, cloud, a_np[picks], task)["state_seqs"]
    err = np.abs(first[picks].cpu().numpy() - want).reshape(len(picks), -1).max(1)
    assert (err <= POS_TOL).sum() >= 2, err                             # (a long free-running rollout may pass a near-tie)


def test_two_contexts_in_one_process_keep_their_own_options(ag, O, dev):
    """Options live in the context (environment defaults are read once, at ag_ctx_create): two models with different
    switches run side by side, each on its own path, and agree bit for bit."""
    rng = np.random.default_rng(99)
    task = _task("rope")
    W, m1 = _model(ag, O, "rope", 99, dev)
    _, m2 = _model(ag, O, "rope", 99, dev)
    cloud = _rope(150, rng)
    reps = rng.integers(1, 6, (40, 1))
    a = torch.from_numpy(_actions(cloud, 40, 1, reps, rng)).to(dev)
    s0 = torch.from_numpy(cloud).to(dev)
    e1, e2 = m1.engine(dev), m2.engine(dev)
    assert e1.ctx.value != e2.ctx.value
    e1.set_option("repeat_sort", 0); e1.set_option("latency", 0); e1.set_option("self_dedupe", 0); e1.set_option("stagger_us", 30)
    assert (e1.get_option("repeat_sort"), e2.get_option("repeat_sort")) == (0, 1)
    assert (e1.get_option("latency"), e2.get_option("latency")) == (0, -1)
    o1 = ag.dynamics(s0, a, m1, dev, _ppm(task, "rope"))["state_seqs"]
    o2 = ag.dynamics(s0, a, m2, dev, _ppm(task, "rope"))["state_seqs"]
    o1b = ag.dynamics(s0, a, m1, dev, _ppm(task, "rope"))["state_seqs"]
    assert torch.equal(o1, o2) and torch.equal(o1, o1b)
    x1, n1 = e1.rollout_counts()
    x2, n2 = e2.rollout_counts()
    assert n1 == n2 == int(reps.sum()) and x2 == n2 and x1 == 40 * int(reps.max())
    with pytest.raises(AssertionError, match="unknown option"):
        e1.set_option("no_such_switch", 1)


def test_shipped_planner_configuration_chunked_equals_loop_at_size(ag, O, dev):
    """The shipped planner workload at its own size (planning/rope.yaml: n_sample_chunk 500, n_look_ahead 1, push length
    U[5,15) -> action_repeat 5..14, 200+1 particles; tools/bench_planner.py times 40 such chunks): two chunks through the
    reference's host loop (plan.py:241-247) and through trajectory_optimization_chunked - bit-identical, repeat-sorted
    launch order included (the 1000-candidate call cuts its launch chunks differently from the 500-candidate calls) - and
    the winner against the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_planner as BP
    rng = np.random.default_rng(0)
    planner, m, s0, lo, hi, cloud, task = BP.make_planner("rope", 500, rng)
    torch.manual_seed(0)
    act_seq = torch.rand((1, 4), device=dev) * (hi - lo) + lo
    torch.manual_seed(1)
    loop = BP.loop_call(planner, s0, act_seq, 2)
    gen = torch.cuda.get_rng_state(dev)
    torch.manual_seed(1)
    fused = planner.trajectory_optimization_chunked(s0, act_seq, 2)
    assert torch.equal(torch.cuda.get_rng_state(dev), gen)
    assert torch.equal(fused["act_seq"], loop["act_seq"])
    assert torch.equal(fused["best_model_output"]["state_seqs"], loop["best_model_output"]["state_seqs"])
    assert torch.equal(fused["best_eval_output"]["reward_seqs"], loop["best_eval_output"]["reward_seqs"])
    # the winners' rollouts sliced out of the big batch instead of rolled out again: the same bits (batch independence)
    planner.reuse_best_rollout = True
    torch.manual_seed(1)
    sliced = planner.trajectory_optimization_chunked(s0, act_seq, 2)
    torch.manual_seed(1)
    sliced_loop = BP.loop_call(planner, s0, act_seq, 2)
    planner.reuse_best_rollout = False
    for r in (sliced, sliced_loop):
        assert torch.equal(r["act_seq"], loop["act_seq"])
        assert torch.equal(r["best_model_output"]["state_seqs"], loop["best_model_output"]["state_seqs"])
        assert torch.equal(r["best_eval_output"]["reward_seqs"], loop["best_eval_output"]["reward_seqs"])
    # the 1000-candidate rollout ran exactly sum(action_repeat) candidate-forwards
    torch.manual_seed(1)
    a = torch.cat([planner.sample_action_sequences(act_seq, iter_index=0) for _ in range(2)])
    planner.model_rollout(s0, a)
    ex, need = m.engine(dev).rollout_counts()
    assert need == int(a[:, 0, 3].to(torch.int32).sum()) and ex <= need    # (ex < need: the contact-free prefix is the base rollout's)
    with m.engine(dev).options(share_prefix=0):
        planner.model_rollout(s0, a)
        assert m.engine(dev).rollout_counts() == (need, need)
    assert 5 <= int(a[:, 0, 3].min()) and int(a[:, 0, 3].max()) <= 14
    W = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = O.dynamics(W, 3, cloud, loop["act_seq"].cpu().numpy()[None], {k: v for k, v in task.items()})["state_seqs"]
    assert np.abs(loop["best_model_output"]["state_seqs"].cpu().numpy() - want).max() <= POS_TOL


# ------------------------------------------------------------------------------------------------- device-planned rollouts
@pytest.mark.parametrize("name,material", [("dyn_rope", "rope"), ("dyn_granular", "granular"), ("dyn_cloth", "cloth")])
def test_device_decoded_actions_vs_reference_golden(ag, dev, name, material):
    """GPU-resident actions + a task config that bounds the push length take ag_rollout_actions: decode_action, the tool
    keypoints (1-point / 5-point pusher, gripper) and the launch plan run in a device kernel.  Against the reference's
    goldens: decoded actions to 1e-6 (device cos/sin), states to 1e-5; the host-decode path stays selectable and bit-equal."""
    from helpers import load_golden, task_of
    from test_gpu_parity import _model as _gmodel
    g = load_golden(name)
    task = task_of(g)
    m = _gmodel(ag, g, material, dev)
    s0, a = torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev)
    host = ag.dynamics(s0, a, m, dev, _ppm(task, material))
    assert torch.equal(host["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    tdev = dict(task, action_upper_lim=[0.0, 4.5, 3.14, 4.0], action_lower_lim=[-4.5, -2.5, -3.14, 2.0])
    eng = m.engine(dev)
    out = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
    assert eng.rollout_counts()[0] == eng.rollout_counts()[1] == int(g["action"][..., 3].astype(np.int32).sum())
    assert out["action_seqs"].is_cuda
    assert float((out["action_seqs"].cpu() - torch.from_numpy(g["action_seqs"])).abs().max()) <= 1e-6
    assert np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).max() <= POS_TOL
    with eng.options(device_decode=0):                                  # same task config, host decode: bit-equal again
        again = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
    assert torch.equal(again["action_seqs"], host["action_seqs"]) and torch.equal(again["state_seqs"], host["state_seqs"])
    cpu_actions = ag.dynamics(s0, a.cpu(), m, dev, _ppm(tdev, material))   # host-resident actions never take the device path
    assert torch.equal(cpu_actions["state_seqs"], host["state_seqs"])


def test_device_planned_rollout_is_invariant_to_order_chunks_and_streams(ag, O, dev):
    """ag_rollout_actions with mixed repeats 2..9 (incl. 0): the device's repeat-sorted plan vs the unsorted one (every slot
    stepped while any is live), one / two streams, odd chunk sizes - identical bits; executed candidate-forwards = sum of
    repeats; a repeat beyond the task config's bound is reported, not silently truncated."""
    rng = np.random.default_rng(103)
    task = _task("cloth", max_nR=40000, action_lower_lim=[-4.5, -2.5, -3.14, 2.0], action_upper_lim=[0.0, 4.5, 3.14, 10.0])
    W, m = _model(ag, O, "cloth", 103, dev)
    cloud = _grid(30, 0.3, 0.02, rng)
    B, H = 150, 2
    reps = rng.integers(2, 10, (B, H))
    reps[5, 1] = 0
    a_np = _actions(cloud, B, H, reps, rng, spread=0.8)
    a_np[5, 1, 3] = 0.25
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, "cloth")
    eng = m.engine(dev)
    outs = []
    for streams, chunk, sort in ((1, 0, 1), (2, 0, 1), (1, 37, 1), (2, 41, 0), (1, 0, 0)):
        with eng.options(streams=streams, repeat_sort=sort):
            eng.set_chunk(chunk)
            try:
                o = ag.dynamics(s0, a, m, dev, ppm)
                ex, need = eng.rollout_counts()
            finally:
                eng.set_chunk(0)
        assert need == int(reps.sum()) and (ex == need if sort else ex > need), (streams, chunk, sort, ex, need)
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(o["state_seqs"], outs[0]["state_seqs"]) and torch.equal(o["action_seqs"], outs[0]["action_seqs"])
    assert float(outs[0]["state_seqs"][5, 1].abs().max()) == 0.0        # repeat 0: the slot stays zero
    with eng.options(device_decode=0):
        host = ag.dynamics(s0, a, m, dev, ppm)
    assert float((host["action_seqs"] - outs[0]["action_seqs"]).abs().max()) <= 1e-6
    err = (host["state_seqs"] - outs[0]["state_seqs"]).abs().reshape(B, -1).max(1).values
    bad = [int(b) for b in torch.nonzero(err > POS_TOL).flatten()]
    assert len(bad) <= 3, err.topk(5)                                   # (1e-7 tool offsets may pass a near-tie in a long rollout)
    for b in bad:   # ... and each of them must BE one: somewhere along its rollout the edge selection is undecided within the tolerance
        tr = []
        O.dynamics(W, 3, cloud, a_np[[b]], task, trace=tr)
        N = cloud.shape[0] + 1
        tool = np.zeros(N, bool)
        tool[-1] = True
        margin = min(O.selection_margin(rec["state_last"], task["adj_thresh"], np.ones(N, bool), tool, task["topk"]) for rec in tr[0])
        assert margin < 4.0 * task["adj_thresh"] * POS_TOL and float(err[b]) < 1e-2, (b, margin, float(err[b]))
    # A repeat beyond the task config's bound.  device_decode = 1: reported, not truncated.  Automatic mode (-1, default): the
    # reference accepts any action length (forward_dynamics.py:156 steps to the batch maximum), so the call is served by the
    # host-decode path instead; a call that does not wait for its result (_sync=False) marks the candidate's rows NaN.
    too_long = a.clone()
    too_long[7, 0, 3] = 12.5
    with eng.options(device_decode=1):
        with pytest.raises(ValueError, match="action_upper_lim"):
            ag.dynamics(s0, too_long, m, dev, ppm)
    ok = ag.dynamics(s0, a, m, dev, ppm)                                 # the context is still usable
    assert torch.equal(ok["state_seqs"], outs[0]["state_seqs"])
    with eng.options(device_decode=0):
        want_long = ag.dynamics(s0, too_long, m, dev, ppm)
    got_long = ag.dynamics(s0, too_long, m, dev, ppm)                    # automatic mode falls back
    assert torch.equal(got_long["state_seqs"], want_long["state_seqs"]) and torch.equal(got_long["action_seqs"], want_long["action_seqs"])
    assert float(got_long["state_seqs"][7, 0].abs().max()) > 0.0
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    o = ag.dynamics(s0, too_long, m, dev, ppm, _sync=False, _overflow_flag=flags)
    torch.cuda.synchronize()
    assert int(flags[1]) == 12 and torch.isnan(o["state_seqs"][7]).all()
    rest = [b for b in range(B) if b != 7]
    assert torch.equal(o["state_seqs"][rest], outs[0]["state_seqs"][rest])
    # a bound the device plan cannot serve: automatic mode takes the host path, device_decode = 1 reports it
    ppm_big = _ppm(dict(task, action_upper_lim=[0.0, 4.5, 3.14, 5000.0]), "cloth")
    assert torch.equal(ag.dynamics(s0, a, m, dev, ppm_big)["state_seqs"], host["state_seqs"])
    with eng.options(device_decode=1):
        with pytest.raises(AssertionError, match="max_repeat"):
            ag.dynamics(s0, a, m, dev, ppm_big)


def test_rollout_with_the_softbody_model_variant_vs_reference_golden(ag, dev):
    """n_his = 5 / pstep = 4 / rel_input_dim 20 (config/dynamics/softbody.yaml) through the rollout driver: dynamics() takes
    n_his from the task config (forward_dynamics.py:16).  Golden from the reference (rope task with n_his 5, the softbody
    model); host-decoded and device-planned actions; teacher-forced edges at every forward."""
    from helpers import load_golden, task_of, split_edges
    g = load_golden("dyn_softbody_nhis5")
    task = task_of(g)
    assert task["n_his"] == 5 and int(g["pstep"]) == 4
    mc, mat, ds = _cfg("softbody", 4)
    ds = dict(ds, n_his=5)
    m = ag.DynamicsPredictor(mc, mat, ds, dev)
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w::")})
    s0, a = torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev)
    out = ag.dynamics(s0, a, m, dev, _ppm(task, "softbody"))
    assert torch.equal(out["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    assert np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).max() <= POS_TOL
    tdev = dict(task, action_upper_lim=[0.0, 4.5, 3.14, 5.0])
    dv = ag.dynamics(s0, a, m, dev, _ppm(tdev, "softbody"))
    assert np.abs(dv["state_seqs"].cpu().numpy() - g["state_seqs"]).max() <= POS_TOL
    sub = ag.dynamics(s0, a[1:2], m, dev, _ppm(task, "softbody"))["state_seqs"]
    assert torch.equal(sub, out["state_seqs"][1:2])
    # the contact-free prefix with five history frames: far pushes appended to the golden's, forced sharing, identical bits
    far = a.clone()
    far[..., 0] += 30.0
    both = torch.cat([a, far, far])
    eng = m.engine(dev)
    with eng.options(share_prefix=1):
        shared = ag.dynamics(s0, both, m, dev, _ppm(task, "softbody"))["state_seqs"]
        ex, need = eng.rollout_counts()
    with eng.options(share_prefix=0):
        plain = ag.dynamics(s0, both, m, dev, _ppm(task, "softbody"))["state_seqs"]
    assert torch.equal(shared, plain) and torch.equal(shared[:len(a)], out["state_seqs"]) and ex < need
    N = g["state0"].shape[0] + 1
    mask = torch.ones((3, N), dtype=torch.bool, device=dev)
    tool = torch.zeros((3, N), dtype=torch.bool, device=dev)
    tool[:, -1] = True
    for i in range(int(g["n_steps"])):
        el = ag.construct_edges_index(torch.from_numpy(g[f"step{i}::state_last"]).to(dev), task["adj_thresh"], mask, tool,
                                      task["topk"], task["connect_tools_all"])
        for (r, s), (wr, ws) in zip(_edges_to_lists(el), split_edges(g, f"step{i}::")):
            assert np.array_equal(r, wr) and np.array_equal(s, ws), i
    with pytest.raises(AssertionError, match="n_his"):
        ag.dynamics(s0, a, m, dev, _ppm(dict(task, n_his=4), "softbody"))


@pytest.mark.parametrize("material,cloud_fn", [("rope", lambda r: _rope(150, r)), ("cloth", lambda r: _grid(16, 0.3, 0.02, r))])
def test_twenty_look_ahead_steps_of_one_repeat_vs_oracle(ag, O, dev, material, cloud_fn):
    """SURVEY 8(d)'s secondary mapping of "horizon 20": n_look_forward = 20, length 1.5 (repeat 1) - the history is reset to the
    captured state at every step (forward_dynamics.py:37-38) and every look-ahead step re-encodes the tool rows.  Host-decoded
    and device-planned actions, against the oracle; the device plan's sorted and unsorted launch orders agree bit for bit."""
    rng = np.random.default_rng(107)
    task = _task(material, max_nR=20000)
    W, m = _model(ag, O, material, 107, dev)
    cloud = cloud_fn(rng)
    B, H = 6, 20
    reps = np.ones((B, H), np.int64)
    reps[2, 5:9] = 2                                                    # a few longer pushes in between
    a_np = _actions(cloud, B, H, reps, rng, spread=0.5)
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    host = ag.dynamics(s0, a, m, dev, _ppm(task, material))
    assert host["state_seqs"].shape == (B, H, cloud.shape[0], 3)
    want = O.dynamics(W, 3, cloud, a_np, task)["state_seqs"]
    err = np.abs(host["state_seqs"].cpu().numpy() - want).reshape(B, -1).max(1)
    assert (err <= POS_TOL).sum() >= B - 1, err
    tdev = dict(task, action_upper_lim=[0.0, 4.5, 3.14, 2.0])
    devp = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
    e2 = (devp["state_seqs"] - host["state_seqs"]).abs().reshape(B, -1).max(1).values
    assert int((e2 <= POS_TOL).sum()) >= B - 1, e2
    eng = m.engine(dev)
    assert eng.rollout_counts() == (int(reps.sum()), int(reps.sum()))
    with eng.options(repeat_sort=0):
        unsorted = ag.dynamics(s0, a, m, dev, _ppm(tdev, material))
    assert torch.equal(unsorted["state_seqs"], devp["state_seqs"])


def test_rollout_graph_with_the_builders_largest_rows(ag, O, dev):
    """The slot-indexed rollout graph at the edge builder's limits: 4000 + 1 particles, top-k 128 (129 slots per row: 516,129 slots,
    a 63-KB bitmap in k_ell_index's LDS - beyond the 64 KB a kernel gets without the opt-in once the static part is added) with a
    radius that keeps ~9 senders per row, and a dense variant (radius 0.26: ~60 senders per row).  Against the oracle."""
    rng = np.random.default_rng(97)
    cloud = _grid(64, 0.1, 0.01, rng)[:4000]
    for thr, tol_scale in ((0.16, 1.0), (0.26, 1.0)):
        task = _task("rope", adj_thresh=thr, topk=128, max_nR=4001 * 129)
        W, m = _model(ag, O, "rope", 97, dev)
        a_np = _actions(cloud, 2, 1, [2, 1], rng, spread=1.0)
        out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev), m, dev, _ppm(task, "rope"))["state_seqs"]
        want = O.dynamics(W, 3, cloud, a_np, task)["state_seqs"]
        assert np.abs(out.cpu().numpy() - want).max() <= POS_TOL * tol_scale, thr
