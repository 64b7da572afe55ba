"""Pin the numpy oracle against outputs of the REAL reference (tests/golden/*.npz, made by
tests/golden/make_golden.py in the build container).  CPU only.

Bars: edge indices bit-exact; positions within POS_TOL per forward and over the free-running rollout.
"""
import numpy as np
import pytest

from oracle import adaptigraph_oracle as O
from helpers import load_golden, task_of, split_edges, golden_step_index

POS_TOL = 5e-6   # fp32; BLAS summation order differs between torch-MKL and numpy (oracle docstring)


def test_edges_batch_bit_exact():
    g = load_golden("edges_batch")
    import json
    cases = json.loads(bytes(g["cases_json"]).decode())
    for ci, c in enumerate(cases):
        pre = f"case{ci}::"
        thr = c["adj_thresh"] if c["adj_thresh"] is not None else g[pre + "adj_thresh_vec"]
        got = O.construct_edges_batch(g[pre + "states"], thr, g[pre + "mask"], g[pre + "tool_mask"],
                                      c["topk"], c["connect_tools_all"], check_ties=True)
        want = split_edges(g, pre)
        for b, ((r, s), (wr, ws)) in enumerate(zip(got, want)):
            assert np.array_equal(r, wr) and np.array_equal(s, ws), (ci, b)


def test_forward_per_particle_physics():
    g = load_golden("forward_perparticle_phys")
    W = O.weights_from_npz(g)
    edges = split_edges(g, "")
    pos, mot = O.model_forward(W, g["state"], g["attrs"], edges, g["p_instance"], g["action"],
                               g["physics_param"], int(g["pstep"]))
    assert np.abs(pos - g["pred_pos"]).max() < POS_TOL
    assert np.abs(mot - g["pred_motion"]).max() < POS_TOL


@pytest.mark.parametrize("name", ["dyn_rope", "dyn_granular", "dyn_cloth", "dyn_softbody_nhis5"])
def test_dynamics_free_running(name):
    g = load_golden(name)
    W, task = O.weights_from_npz(g), task_of(g)
    trace = []
    out = O.dynamics(W, int(g["pstep"]), g["state0"], g["action"], task, trace=trace)
    assert np.array_equal(out["action_seqs"], g["action_seqs"])
    _, rep = O.decode_action(g["action"], task["push_length"])
    base = golden_step_index(rep)
    for b, tr in enumerate(trace):
        k = 0
        for li in range(rep.shape[1]):
            for ai in range(rep[b, li]):
                gi = base[li] + ai
                wr, ws = split_edges(g, f"step{gi}::")[b]
                assert np.array_equal(tr[k]["recv"], wr) and np.array_equal(tr[k]["send"], ws), (b, li, ai)
                assert np.abs(tr[k]["pred_pos"] - g[f"step{gi}::pred_pos"][b]).max() < POS_TOL
                k += 1
    assert np.abs(out["state_seqs"] - g["state_seqs"]).max() < POS_TOL


@pytest.mark.parametrize("name", ["dyn_masked_rope", "dyn_masked_cloth", "dyn_masked_granular"])
def test_dynamics_masked(name):
    g = load_golden(name)
    W, task = O.weights_from_npz(g), task_of(g)
    out = O.dynamics_masked(W, int(g["pstep"]), g["state_init"], g["state_mask"], g["action"], task)
    assert np.array_equal(out["action_seqs"], g["action_seqs"])
    assert np.abs(out["state_seqs"] - g["state_seqs"]).max() < POS_TOL


def test_overflow_raises_like_reference():
    g = load_golden("dyn_overflow")
    W, task = O.weights_from_npz(g), task_of(g)
    with pytest.raises(Exception, match="Exceeds max dims"):
        O.dynamics(W, int(g["pstep"]), g["state0"], g["action"], task)
    assert bytes(g["expected_exception"]).decode() == "Exceeds max dims"


# ------------------------------------------------------------------------------------------------- cost functions
COST_TOL = 2e-5


def test_costs_losses_vs_reference():
    from functools import partial
    from oracle import costs_oracle as C
    g = load_golden("costs")
    B, H, N, _ = g["state"].shape
    flat = g["state"].reshape(B * H, N, 3)
    assert np.abs(C.chamfer(flat, g["target"][None]) - g["chamfer"]).max() < COST_TOL
    assert np.abs(C.box_loss(flat, g["target_box"]) - g["box_loss"]).max() < COST_TOL
    for kind in ("rope", "cloth", "granular"):
        got = getattr(C, kind + "_penalty")(g["state"], g["action"], g["state_cur"], 10.0)
        assert np.abs(got - g[kind + "_penalty"]).max() < COST_TOL, kind
    mc = C.mean_chamfer(g["mc_pred"], g["mc_real"], g["mc_pred_mask"], g["mc_real_mask"])
    assert np.abs(mc - g["mean_chamfer"]).max() < COST_TOL
    for err_name in ("chamfer", "box"):
        err = partial(C.chamfer, y=g["target"][None]) if err_name == "chamfer" else partial(C.box_loss, target=g["target_box"])
        for kind in ("rope", "cloth", "granular"):
            pen = partial(getattr(C, kind + "_penalty"), sim_real_ratio=10.0)
            r = C.running_cost(g["state"], g["action"], g["state_cur"], err, pen, g["bbox"])
            want = g[f"reward::{err_name}::{kind}"]
            assert np.abs(r - want).max() < 5e-5 * max(1.0, np.abs(want).max()), (err_name, kind)


def test_ppm_dynamics_error_vs_reference():
    from oracle import costs_oracle as C
    g = load_golden("ppm_dynamics_error")
    W, task = O.weights_from_npz(g), task_of(g)
    n = int(g["n_act"])
    inits, reals, acts = ([g[f"{k}{i}"] for i in range(n)] for k in ("init", "real", "act"))
    for v, want in zip(g["phys_values"], g["errors"]):
        got = C.dynamics_error(W, int(g["pstep"]), float(v), task, inits, reals, acts)
        assert abs(got - want) < 2e-5, (v, got, want)


def test_single_graph_builder_vs_reference():
    import json
    g = load_golden("edges_single")
    cases = json.loads(bytes(g["cases_json"]).decode())
    for ci, c in enumerate(cases):
        pre = f"case{ci}::"
        r, s = O.construct_edges_from_states(g[pre + "states"], c["adj_thresh"], g[pre + "mask"], g[pre + "tool_mask"],
                                             c["topk"], c["connect_tools_all"])
        (wr, ws), = split_edges(g, pre)
        assert np.array_equal(r, wr) and np.array_equal(s, ws), ci


def _rule_kwargs(c):
    kw = {k: np.float32(v) for k, v in c["bounds_f32"].items()}        # numpy float32 scalars, as the eval rollout passes
    kw.update(topk=c["topk"], connect_tools_all=c["connect_tools_all"], connect_tools_surface=c["connect_tools_surface"],
              connect_tool_all_non_fixed=c["connect_tool_all_non_fixed"], kNN=c["kNN"])
    return kw


def test_single_graph_tool_rules_vs_reference():
    """graph.py:125-221: non-fixed-particle rule, flat kNN filter, two-closest-surface-planes rule, and both in sequence."""
    import json
    g = load_golden("edges_single_rules")
    meta = json.loads(bytes(g["meta_json"]).decode())
    seen = set()
    for ci, c in enumerate(meta["cases"]):
        pre = f"case{ci}::"
        tr = {}
        r, s = O.construct_edges_from_states(g[pre + "states"], c["adj_thresh"], g[pre + "mask"], g[pre + "tool_mask"],
                                             trace=tr, **_rule_kwargs(c))
        (wr, ws), = split_edges(g, pre)
        assert np.array_equal(r, wr) and np.array_equal(s, ws), (ci, tr)
        seen |= set(tr)
    assert {"nonfixed_check", "keepK", "surface_check", "planes"} <= seen


def test_backoff_loop_vs_reference():
    """rollout.py:185-222 replayed with the oracle builder: same trail of (kNN, topk, n_rel), same final graph."""
    import json
    g = load_golden("edges_single_rules")
    b = json.loads(bytes(g["meta_json"]).decode())["backoff"]
    args = (g["backoff::states"], b["adj_thresh"], g["backoff::mask"], g["backoff::tool_mask"])
    bounds = dict(max_y=np.float32(b["max_y"]), min_y=np.float32(b["min_y"]))
    kNN, dec, trail = b["knn_thresh"], b["topk"], []
    r, s = O.construct_edges_from_states(*args, topk=b["topk"], kNN=kNN, **bounds)
    assert len(r) == b["first_n_rel"]
    while len(r) > b["max_nR"]:
        if kNN <= b["min_kNN"]:
            dec -= 1
            r, s = O.construct_edges_from_states(*args, topk=dec, kNN=kNN, **bounds)
        else:
            kNN = kNN - b["knn_increment"]
            r, s = O.construct_edges_from_states(*args, topk=b["topk"], kNN=kNN, **bounds)
        trail.append([float(kNN), int(dec), len(r)])
    assert trail == b["trail"]
    (wr, ws), = split_edges(g, "backoff::")
    assert np.array_equal(r, wr) and np.array_equal(s, ws)


def test_dynamics_repeat_zero_vs_reference():
    """forward_dynamics.py:32,38: action_repeat == 0 leaves that slot zero and the next look-ahead step starts from the
    zero cloud (every top-k choice there is a tie between coincident particles: only the states are compared)."""
    g = load_golden("dyn_rope_repeat0")
    W, task = O.weights_from_npz(g), task_of(g)
    out = O.dynamics(W, int(g["pstep"]), g["state0"], g["action"], task)
    assert np.array_equal(out["action_seqs"], g["action_seqs"])
    assert np.all(g["state_seqs"][0, 0] == 0) and np.all(g["state_seqs"][1, 1] == 0)
    assert np.abs(out["state_seqs"] - g["state_seqs"]).max() < POS_TOL


def test_forward_softbody_variant_nhis5_pstep4():
    """The n_his = 5 / rel_input_dim = 20 / pstep = 4 model of config/dynamics/softbody.yaml (forward only)."""
    g = load_golden("forward_softbody_nhis5")
    W = O.weights_from_npz(g)
    assert W["relation_encoder.model.0.weight"].shape == (150, 20) and g["state"].shape[1] == 5
    edges = split_edges(g, "")
    pos, mot = O.model_forward(W, g["state"], g["attrs"], edges, g["p_instance"], g["action"],
                               g["physics_param"], int(g["pstep"]))
    assert np.abs(pos - g["pred_pos"]).max() < POS_TOL
    assert np.abs(mot - g["pred_motion"]).max() < POS_TOL


def _eval_rollout_fixture(name="eval_rollout_softbody"):
    import json
    g = load_golden(name)
    meta = json.loads(bytes(g["meta_json"]).decode())
    (r0, s0), = split_edges(g, "first::")
    graph = {"state": g["hist0"], "action": g["action0"], "attrs": g["attrs"], "edges": (r0, s0), "p_instance": g["p_instance"],
             "physics": g["physics_param"], "obj_mask": g["obj_mask"], "state_mask": g["state_mask"], "eef_mask": g["eef_mask"]}
    return g, meta, graph


@pytest.mark.parametrize("name,kinds_want", [("eval_rollout_softbody", {"fits", "knn", "topk"}), ("eval_rollout_surface", {"fits", "topk"})])
def test_eval_open_loop_rollout_vs_reference(name, kinds_want):
    """rollout.py:108-260 (softbody.yaml: n_his 5, pstep 4, store_rest_state, tool-to-non-fixed rule with kNN back-off) driven
    step by step: per step the oracle, fed the reference's own graph, reproduces the prediction (tolerance), and - fed the
    reference's prediction - the bounds, the back-off trail and the final edge list (bit-exact); free-running it stays on the
    reference's states over all 12 steps, including the steps whose graph came out of the kNN and the top-k back-off."""
    g, meta, graph = _eval_rollout_fixture(name)
    W, S = O.weights_from_npz(g), g["pred_pos"].shape[0]
    edges = split_edges(g, "step::")
    kinds = set()
    free = dict(graph)
    worst = 0.0
    for i in range(S):
        # (a) teacher-forced: the reference's graph in, its prediction out; its prediction in, its next graph out
        tr = []
        nxt, pred, mot = O.eval_rollout_step(W, 4, graph, g["eef_start"][i], g["eef_end"][i], meta, trail=tr)
        assert np.abs(pred - g["pred_pos"][i]).max() < POS_TOL and np.abs(mot - g["pred_motion"][i]).max() < POS_TOL, i
        b = O.surface_bounds(g["pred_pos"][i][g["obj_mask"]], meta["connect_tool_surface_ratio"])
        assert {k: float(v) for k, v in b.items()} == meta["bounds_f32"][i]
        tr = []
        r, s = O.edges_with_backoff(g["builder_states"][i], meta, g["state_mask"], g["eef_mask"], b, trail=tr)
        assert tr == meta["trails"][i], (i, tr, meta["trails"][i])
        assert np.array_equal(r, edges[i][0]) and np.array_equal(s, edges[i][1]), i
        kinds.add("fits" if len(tr) == 1 else "topk" if tr[-1][1] < meta["topk"] else "knn")
        assert np.array_equal(np.concatenate([g["pred_pos"][i], g["eef_start"][i]], 0), g["builder_states"][i])
        # the reference's next graph (store_rest_state: frame 0 stays, rollout.py:224-229; else a plain shift, :231-232)
        if meta["store_rest_state"]:
            hist = np.concatenate([graph["state"][:1], graph["state"][2:], g["builder_states"][i][None]], 0)
        else:
            hist = np.concatenate([graph["state"][1:], g["builder_states"][i][None]], 0)
        delta = np.zeros_like(g["builder_states"][i]); delta[meta["max_nobj"]:] = g["eef_end"][i] - g["eef_start"][i]
        graph = dict(graph, state=hist, action=delta, edges=edges[i])
        assert np.array_equal(graph["state"][0], g["hist0"][0]) == bool(meta["store_rest_state"] or i < 0)
        # (b) free-running
        free, fpred, _ = O.eval_rollout_step(W, 4, free, g["eef_start"][i], g["eef_end"][i], meta)
        worst = max(worst, float(np.abs(fpred - g["pred_pos"][i]).max()))
        assert np.array_equal(free["edges"][0], edges[i][0]) and np.array_equal(free["edges"][1], edges[i][1]), i
    assert kinds == kinds_want
    assert worst < 1e-5, worst
