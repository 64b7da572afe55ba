"""SURVEY 8(f) rank 3, the step loop: adaptigraph_amd.rollout_eval_step against tests/golden/eval_rollout_softbody.npz - 12 steps of
the reference's eval open-loop rollout (src/dynamics/rollout/rollout.py:108-260; softbody.yaml: n_his 5, pstep 4, store_rest_state,
tool-to-all-non-fixed rule with the kNN / top-k max_nR back-off), recorded by tests/golden/make_golden.py --eval-rollout from the
reference's own model / truncate_graph / construct_edges_from_states / pad_torch calls.  (-m gpu; the oracle's pin on the same file:
tests/test_oracle_vs_golden.py::test_eval_open_loop_rollout_vs_reference.)"""
import json

import numpy as np
import pytest
import torch

from helpers import load_golden, split_edges
from test_gpu_parity import _cfg, POS_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ag():
    import adaptigraph_amd
    return adaptigraph_amd


def _dense(r, s, N, rows, dev):
    Rr, Rs = torch.zeros((1, rows, N)), torch.zeros((1, rows, N))
    Rr[0, torch.arange(len(r)), torch.from_numpy(r).long()] = 1
    Rs[0, torch.arange(len(s)), torch.from_numpy(s).long()] = 1
    return Rr.to(dev), Rs.to(dev)


FIXTURES = [("eval_rollout_softbody", {"fits", "knn", "topk"}),      # softbody.yaml as shipped
            ("eval_rollout_surface", {"fits", "topk"})]               # surface-plane rule on (ratio 0.8), no rest frame, kNN range [1, 1]


def _setup(ag, dev, name="eval_rollout_softbody"):
    g = load_golden(name)
    meta = json.loads(bytes(g["meta_json"]).decode())
    mc, mat, ds = _cfg("softbody", int(g["pstep"]))
    m = ag.DynamicsPredictor(mc, mat, dict(ds, n_his=5), dev)
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w::")})
    N = g["attrs"].shape[0]
    (r0, s0), = split_edges(g, "first::")
    Rr, Rs = _dense(r0, s0, N, meta["max_nR"], dev)                       # padded to max_nR, as rollout/graph.py:508-543 leaves it
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    graph = {"state": t(g["hist0"])[None], "action": t(g["action0"])[None], "Rr": Rr, "Rs": Rs, "attrs": t(g["attrs"])[None],
             "p_rigid": torch.zeros(1, 1, device=dev), "p_instance": t(g["p_instance"])[None], "obj_mask": t(g["obj_mask"])[None],
             "eef_mask": t(g["eef_mask"])[None], "state_mask": t(g["state_mask"])[None],
             "material_index": torch.ones(1, g["p_instance"].shape[0], 1, dtype=torch.long, device=dev),
             "softbody_physics_param": t(g["physics_param"])[None]}
    kw = dict(adj_thresh=meta["adj_thresh"], topk=meta["topk"], max_nR=meta["max_nR"], connect_tool_all=meta["connect_tool_all"],
              connect_tool_all_non_fixed=meta["connect_tool_all_non_fixed"], connect_tool_surface=meta["connect_tool_surface"],
              connect_tool_surface_ratio=meta["connect_tool_surface_ratio"], knn_thresh=meta["knn_thresh"], min_kNN=meta["min_kNN"],
              knn_increment=meta["knn_increment"], store_rest_state=meta["store_rest_state"])
    return g, meta, m, graph, kw


@pytest.mark.parametrize("name,kinds_want", FIXTURES)
def test_backoff_rebuild_on_the_references_predictions_is_bit_exact(ag, dev, name, kinds_want):
    """Teacher-forced: the cloud the reference's builder was fed at every step -> the same bounds, the same back-off trail
    (kNN 0.7 -> 0.4 in steps of 0.1, then top-k 10 -> 9) and the same final edge list, bit for bit."""
    g, meta, m, graph, kw = _setup(ag, dev, name)
    edges = split_edges(g, "step::")
    kinds = set()
    for i in range(g["pred_pos"].shape[0]):
        pred = torch.from_numpy(g["pred_pos"][i]).to(dev)
        b = ag.surface_bounds(pred[torch.from_numpy(g["obj_mask"]).to(dev)], meta["connect_tool_surface_ratio"])
        assert {k: float(v) for k, v in b.items()} == meta["bounds_f32"][i], i
        tr = []
        el = ag.construct_edges_with_backoff(torch.from_numpy(g["builder_states"][i]).to(dev), meta["adj_thresh"],
                                             torch.from_numpy(g["state_mask"]).to(dev), torch.from_numpy(g["eef_mask"]).to(dev), meta["topk"],
                                             meta["max_nR"], knn_thresh=meta["knn_thresh"], min_kNN=meta["min_kNN"],
                                             knn_increment=meta["knn_increment"], as_index=True, trail=tr,
                                             connect_tools_all=meta["connect_tool_all"], connect_tools_surface=meta["connect_tool_surface"],
                                             connect_tool_all_non_fixed=meta["connect_tool_all_non_fixed"], **b)
        assert [list(x) for x in tr] == meta["trails"][i], (i, tr)
        n = int(el.n_edges[0])
        assert np.array_equal(el.recv[0, :n].cpu().numpy(), edges[i][0]) and np.array_equal(el.send[0, :n].cpu().numpy(), edges[i][1]), i
        kinds.add("fits" if len(tr) == 1 else "topk" if tr[-1][1] < meta["topk"] else "knn")
    assert kinds == kinds_want


@pytest.mark.parametrize("name", [f[0] for f in FIXTURES])
@pytest.mark.parametrize("dense", [True, False])
def test_free_running_eval_rollout_stays_on_the_references_states(ag, dev, dense, name):
    """12 steps free-running through rollout_eval_step (dense=True: the graph dictionary carries one-hot Rr / Rs padded to max_nR like
    the reference's, and `truncate_graph` + `model(**graph)` work on it; dense=False: index lists): predictions within 1e-5 of the
    reference's at every step, the rebuilt graphs identical to its (back-off steps included), the rest frame kept in slot 0 (or, without
    store_rest_state, the history shifted by one)."""
    g, meta, m, graph, kw = _setup(ag, dev, name)
    edges = split_edges(g, "step::")
    worst = 0.0
    for i in range(g["pred_pos"].shape[0]):
        tr = []
        graph, pred, mot = ag.rollout_eval_step(m, graph, g["eef_start"][i], g["eef_end"][i], dense=dense, trail=tr, **kw)
        err = float(np.abs(pred[0].cpu().numpy() - g["pred_pos"][i]).max())
        worst = max(worst, err)
        assert err <= POS_TOL, (i, err)
        assert float(np.abs(mot[0].cpu().numpy() - g["pred_motion"][i]).max()) <= POS_TOL
        assert [list(x) for x in tr] == meta["trails"][i], (i, tr)
        if dense:
            assert graph["Rr"].shape == (1, meta["max_nR"], g["attrs"].shape[0])
            el = ag.EdgeList.from_dense(graph["Rr"], graph["Rs"])
            assert ag.truncate_graph(dict(graph))["Rr"].shape[1] == len(edges[i][0])        # utils.py:150-160 on the engine's dictionary
        else:
            el = graph["edges"]
        n = int(el.n_edges[0])
        assert np.array_equal(el.recv[0, :n].cpu().numpy(), edges[i][0]) and np.array_equal(el.send[0, :n].cpu().numpy(), edges[i][1]), i
        if meta["store_rest_state"]:
            assert torch.equal(graph["state"][0, 0].cpu(), torch.from_numpy(g["hist0"][0]))  # rollout.py:224-229
        assert torch.equal(graph["state"][0, -1, :g["pred_pos"].shape[1]], pred[0])
        assert np.array_equal(graph["action"][0, meta["max_nobj"]:].cpu().numpy(), g["eef_end"][i] - g["eef_start"][i])
    print(f"{name} ({'dense Rr/Rs' if dense else 'index lists'}): max-abs position error over {g['pred_pos'].shape[0]} free-running steps {worst:.2e}")


def test_without_store_rest_state_the_history_shifts_by_one(ag, dev):
    g, meta, m, graph, kw = _setup(ag, dev)
    before = graph["state"].clone()
    nxt, pred, _ = ag.rollout_eval_step(m, graph, g["eef_start"][0], g["eef_end"][0], dense=False, **dict(kw, store_rest_state=False))
    assert torch.equal(nxt["state"][0, :-1], before[0, 1:])                                  # rollout.py:231-232
    assert torch.equal(nxt["state"][0, -1, :pred.shape[1]], pred[0])


def test_multi_step_driver_equals_the_steps(ag, dev):
    """rollout_eval = the step function along the tool trajectory (frame pairs as rollout.py:158-161 looks them up): same predictions
    and trails as stepping by hand, bit for bit; prints the time per step beside the reference's (this container, 8 threads)."""
    import time
    g, meta, m, graph, kw = _setup(ag, dev)
    S, n_his = g["pred_pos"].shape[0], int(g["n_his"])
    by_hand, gr = [], dict(graph)
    for i in range(S):
        gr, pred, _ = ag.rollout_eval_step(m, gr, g["eef_start"][i], g["eef_end"][i], dense=False, **kw)
        by_hand.append(pred[0].clone())
    assert np.array_equal(g["eef_pos"][n_his], g["eef_start"][0]) and np.array_equal(g["eef_pos"][n_his + 1], g["eef_end"][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, preds, trails = ag.rollout_eval(m, graph, torch.from_numpy(g["eef_pos"]), n_his, S, dense=False, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / S
    assert all(torch.equal(a, b) for a, b in zip(preds, by_hand))
    assert [[list(x) for x in t] for t in trails] == meta["trails"]
    print(f"eval open-loop step (230 + 5 particles, back-off included): {dt * 1e3:.2f} ms per step on the engine")
