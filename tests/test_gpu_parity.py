"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against
  (1) golden vectors produced by the REAL reference (tests/golden/*.npz), and
  (2) the numpy oracle (oracle/) on seeded inputs.
Bars: edge indices bit-exact; particle positions max-abs error <= 1e-5 (BASELINE.json north_star).
"""
import json
import types

import numpy as np
import pytest
import torch

from helpers import load_golden, task_of, split_edges, golden_step_index

pytestmark = pytest.mark.gpu

POS_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ag():
    import adaptigraph_amd
    return adaptigraph_amd


def _cfg(material, pstep=3):
    phys = {"rope": "particle_radius", "granular": "granular_scale", "cloth": "sf", "softbody": "stiffness"}[material]
    model_config = dict(verbose=False, nf_particle=150, nf_relation=150, nf_effect=150, nf_physics=10, attr_dim=2,
                        state_dim=0, offset_dim=0, action_dim=3, density_dim=0, pstep=pstep, sequence_len=4,
                        rel_particle_dim=0, rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, rel_density_dim=0)
    material_config = {"material_index": {material: 0},
                       material: {"physics_params": [{"name": phys, "use": True, "min": 0.0, "max": 1.0}]}}
    dataset_config = {"n_his": 4, "materials": [material]}
    return model_config, material_config, dataset_config


def _model(ag, g, material, dev):
    m = ag.DynamicsPredictor(*_cfg(material, int(g["pstep"])), dev)
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w::")}
    m.load_state_dict(sd)
    return m.to(dev).eval()


def _ppm(task, material):
    return types.SimpleNamespace(task_config=task, eef_num=task["eef_num"], material=material,
                                 material_dims=task["material_dims"], material_indices=task["material_indices"],
                                 physics_param={material: torch.tensor([0.5])}, adj_thresh=task["adj_thresh"])


def _edges_to_lists(el):
    n = el.n_edges.cpu().numpy()
    r, s = el.recv.cpu().numpy(), el.send.cpu().numpy()
    return [(r[b, :n[b]], s[b, :n[b]]) for b in range(len(n))]


# ------------------------------------------------------------------------------------------------- edges
@pytest.fixture(params=["auto", "rows", "blocks"])
def edge_path(request, ag, dev):
    """The top-k builder has two row schedules with identical results: one receiver row per wavefront, and 64 rows per
    wavefront (taken for slices of >= 256 rows).  Option edge_block_min forces either one at any slice size."""
    if request.param == "auto":
        yield request.param
        return
    with ag.default_engine(dev).options(edge_block_min=1 if request.param == "blocks" else 1000000000):
        yield request.param


def test_edges_vs_reference_golden(ag, dev, edge_path):
    g = load_golden("edges_batch")
    cases = json.loads(bytes(g["cases_json"]).decode())
    for ci, c in enumerate(cases):
        pre = f"case{ci}::"
        thr = c["adj_thresh"] if c["adj_thresh"] is not None else torch.from_numpy(g[pre + "adj_thresh_vec"]).to(dev)
        el = ag.construct_edges_index(torch.from_numpy(g[pre + "states"]).to(dev), thr,
                                      torch.from_numpy(g[pre + "mask"]).to(dev),
                                      torch.from_numpy(g[pre + "tool_mask"]).to(dev), c["topk"], c["connect_tools_all"])
        want = split_edges(g, pre)
        for b, ((r, s), (wr, ws)) in enumerate(zip(_edges_to_lists(el), want)):
            assert np.array_equal(r, wr) and np.array_equal(s, ws), (ci, b)
        # CSR offsets are consistent with recv
        rp = el.row_ptr.cpu().numpy()
        for b, (wr, _) in enumerate(want):
            assert np.array_equal(rp[b], np.concatenate([[0], np.cumsum(np.bincount(wr, minlength=el.N))]))


def test_dense_dropin_matches_reference_layout(ag, dev):
    g = load_golden("edges_batch")
    pre = "case2::"
    Rr, Rs = ag.construct_edges_from_states_batch(torch.from_numpy(g[pre + "states"]).to(dev), 0.4,
                                                  torch.from_numpy(g[pre + "mask"]).to(dev),
                                                  torch.from_numpy(g[pre + "tool_mask"]).to(dev), topk=20,
                                                  connect_tools_all=True)
    want = split_edges(g, pre)
    n_rel = max(len(r) for r, _ in want)
    assert Rr.shape == (3, n_rel, 205) and Rs.shape == Rr.shape
    for b, (wr, ws) in enumerate(want):
        assert np.array_equal(Rr[b, :len(wr)].argmax(-1).cpu().numpy(), wr)
        assert np.array_equal(Rs[b, :len(ws)].argmax(-1).cpu().numpy(), ws)
        assert float(Rr[b, len(wr):].abs().sum()) == 0.0 and float(Rr[b].sum()) == len(wr)
    # and back again through the compat path
    el = ag.EdgeList.from_dense(Rr, Rs)
    for b, ((r, s), (wr, ws)) in enumerate(zip(_edges_to_lists(el), want)):
        assert np.array_equal(r, wr) and np.array_equal(s, ws)


@pytest.mark.parametrize("N_o,M,topk,thr,cta,seed", [
    (700, 1, 7, 10.0, False, 0),      # every sender in radius: streaming top-k compaction (LDS buffer overflows)
    (700, 3, 128, 10.0, True, 1),     # largest supported k < N
    (333, 2, 1000, 0.3, False, 2),    # topk >= N: radius only
    (2025, 1, 5, 0.75, True, 3),      # cloth config-4 size
    (1024, 5, 20, 0.40, False, 4),    # granular config-3 size
    (65, 1, 10, 0.5, False, 5),
    (1, 1, 10, 0.5, True, 6),         # degenerate: single object particle
    (4094, 2, 10, 0.12, True, 7),     # the builder's maximum: 4096 particles (64 sender chunks)
])
def test_edges_vs_oracle(ag, dev, edge_path, N_o, M, topk, thr, cta, seed):
    from oracle import adaptigraph_oracle as O
    rng = np.random.default_rng(seed)
    B, N = 3, N_o + M
    side = int(np.ceil(np.sqrt(N_o)))
    pitch = {0.75: 0.3, 0.40: 0.12}.get(thr, 0.05)
    states = np.zeros((B, N, 3), np.float32)
    mask = np.ones((B, N), bool)
    tool = np.zeros((B, N), bool)
    tool[:, N_o:] = True
    for b in range(B):
        gx = (np.arange(side) * pitch)
        xx, zz = np.meshgrid(gx, gx, indexing="ij")
        p = np.stack([xx.ravel(), np.zeros(side * side), zz.ravel()], 1)[:N_o]
        states[b, :N_o] = p + rng.normal(0, pitch / 6, p.shape)
        states[b, N_o:] = states[b, rng.integers(0, N_o, M)] + rng.normal(0, pitch / 3, (M, 3))
        if b == 1 and N_o > 4:
            mask[b, N_o * 2 // 3:N_o] = False      # ragged batch element
        if b == 2:
            states[b, N_o:, 0] += 1e3 if thr < 5 else 0.0   # tool far away: connect_tools_all flag stays false
    want = O.construct_edges_batch(states, thr, mask, tool, topk, cta, check_ties=True)
    el = ag.construct_edges_index(torch.from_numpy(states).to(dev), thr, torch.from_numpy(mask).to(dev),
                                  torch.from_numpy(tool).to(dev), topk, cta)
    for b, ((r, s), (wr, ws)) in enumerate(zip(_edges_to_lists(el), want)):
        assert len(r) == len(wr), (b, len(r), len(wr))
        assert np.array_equal(r, wr) and np.array_equal(s, ws), b


def test_edges_exact_ties_use_distance_then_index(ag, dev, edge_path):
    """Regular lattice: mass ties at the k-th boundary.  The build's documented rule is (distance, index)."""
    from oracle import adaptigraph_oracle as O
    g = (np.arange(12) * 0.25).astype(np.float32)
    xx, zz = np.meshgrid(g, g, indexing="ij")
    states = np.stack([xx.ravel(), np.zeros(144, np.float32), zz.ravel()], 1)[None].astype(np.float32)
    mask = np.ones((1, 144), bool)
    tool = np.zeros((1, 144), bool)
    want = O.construct_edges_batch(states, 0.6, mask, tool, 4, False)
    el = ag.construct_edges_index(torch.from_numpy(states).to(dev), 0.6, torch.from_numpy(mask).to(dev),
                                  torch.from_numpy(tool).to(dev), 4, False)
    (r, s), (wr, ws) = _edges_to_lists(el)[0], want[0]
    assert np.array_equal(r, wr) and np.array_equal(s, ws)


# ------------------------------------------------------------------------------------------------- model forward
def test_forward_vs_reference_golden(ag, dev):
    g = load_golden("forward_perparticle_phys")
    m = _model(ag, g, "granular", dev)
    edges = split_edges(g, "")
    B, N = g["attrs"].shape[:2]
    E = max(len(r) for r, _ in edges)
    Rr = torch.zeros((B, E + 5, N))       # padded like pad_torch would
    Rs = torch.zeros((B, E + 5, N))
    for b, (r, s) in enumerate(edges):
        Rr[b, torch.arange(len(r)), torch.from_numpy(r).long()] = 1
        Rs[b, torch.arange(len(s)), torch.from_numpy(s).long()] = 1
    kw = dict(state=torch.from_numpy(g["state"]).to(dev), attrs=torch.from_numpy(g["attrs"]).to(dev),
              p_instance=torch.from_numpy(g["p_instance"]).to(dev), action=torch.from_numpy(g["action"]).to(dev),
              granular_physics_param=torch.from_numpy(g["physics_param"]).to(dev))
    pos, mot = m(Rr=Rr.to(dev), Rs=Rs.to(dev), **kw)
    assert np.abs(pos.cpu().numpy() - g["pred_pos"]).max() <= POS_TOL
    assert np.abs(mot.cpu().numpy() - g["pred_motion"]).max() <= POS_TOL
    # index-list fast path gives the identical bits
    el = ag.construct_edges_index(kw["state"][:, -1], 0.4, torch.ones((B, N), dtype=torch.bool, device=dev),
                                  torch.from_numpy(np.arange(N) >= 150)[None].repeat(B, 1).to(dev), 20, False)
    pos2, mot2 = m(edges=el, **kw)
    assert torch.equal(pos, pos2) and torch.equal(mot, mot2)


def test_forward_vs_oracle_pstep4(ag, dev):
    """softbody.yaml uses pstep=4: no golden, oracle only."""
    from oracle import adaptigraph_oracle as O
    rng = np.random.default_rng(9)
    W = O.random_weights(9)
    m = ag.DynamicsPredictor(*_cfg("rope", 4), dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    B, N_o, M = 2, 97, 1
    N = N_o + M
    state = rng.normal(0, 0.3, (B, 4, N, 3)).astype(np.float32)
    state[:, 1:] = state[:, :1] + rng.normal(0, 0.01, (B, 3, N, 3)).astype(np.float32)
    attrs = np.zeros((B, N, 2), np.float32)
    attrs[:, :N_o, 0] = 1
    attrs[:, N_o:, 1] = 1
    action = np.zeros((B, N, 3), np.float32)
    action[:, N_o:] = 0.05
    p_inst = np.ones((B, N_o, 1), np.float32)
    phys = np.full((B, 1), 0.3, np.float32)
    mask = np.ones((B, N), bool)
    tool = np.zeros((B, N), bool)
    tool[:, N_o:] = True
    edges = O.construct_edges_batch(state[:, -1], 0.5, mask, tool, 10, False, check_ties=True)
    want_pos, want_mot = O.model_forward(W, state, attrs, edges, p_inst, action, phys, 4)
    el = ag.construct_edges_index(torch.from_numpy(state[:, -1]).to(dev), 0.5, torch.from_numpy(mask).to(dev),
                                  torch.from_numpy(tool).to(dev), 10, False)
    pos, mot = m(state=torch.from_numpy(state).to(dev), attrs=torch.from_numpy(attrs).to(dev), edges=el,
                 p_instance=torch.from_numpy(p_inst).to(dev), action=torch.from_numpy(action).to(dev),
                 rope_physics_param=torch.from_numpy(phys).to(dev))
    assert np.abs(pos.cpu().numpy() - want_pos).max() <= POS_TOL
    assert np.abs(mot.cpu().numpy() - want_mot).max() <= POS_TOL


# ------------------------------------------------------------------------------------------------- rollout
@pytest.mark.parametrize("name,material", [("dyn_rope", "rope"), ("dyn_granular", "granular"), ("dyn_cloth", "cloth")])
def test_dynamics_vs_reference_golden(ag, dev, name, material):
    g = load_golden(name)
    task = task_of(g)
    m = _model(ag, g, material, dev)
    ppm = _ppm(task, material)
    out = ag.dynamics(torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev), m, dev, ppm)
    assert torch.equal(out["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    err = np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).max()
    assert err <= POS_TOL, err
    # teacher-forced: rebuild the graph from the positions the reference fed its edge builder at every step
    N = g["state0"].shape[0] + task["eef_num"]
    B = g["action"].shape[0]
    mask = torch.ones((B, N), dtype=torch.bool, device=dev)
    tool = torch.zeros((B, N), dtype=torch.bool, device=dev)
    tool[:, g["state0"].shape[0]:] = True
    for i in range(int(g["n_steps"])):
        el = ag.construct_edges_index(torch.from_numpy(g[f"step{i}::state_last"]).to(dev), task["adj_thresh"], mask,
                                      tool, task["topk"], task["connect_tools_all"])
        for b, ((r, s), (wr, ws)) in enumerate(zip(_edges_to_lists(el), split_edges(g, f"step{i}::"))):
            assert np.array_equal(r, wr) and np.array_equal(s, ws), (i, b)


def test_dynamics_chunking_is_bit_invariant(ag, dev):
    g = load_golden("dyn_rope")
    task = task_of(g)
    m = _model(ag, g, "rope", dev)
    ppm = _ppm(task, "rope")
    s0, a = torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev)
    ref = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    for chunk in (1, 3):
        m.engine(dev).set_chunk(chunk)
        assert torch.equal(ag.dynamics(s0, a, m, dev, ppm)["state_seqs"], ref)
    m.engine(dev).set_chunk(0)
    # a shard of the batch == the same rows of the full batch (what multi-GPU sharding relies on)
    part = ag.dynamics(s0, a[1:3], m, dev, ppm)["state_seqs"]
    assert torch.equal(part, ref[1:3])


@pytest.mark.parametrize("name,material", [("dyn_masked_rope", "rope"), ("dyn_masked_cloth", "cloth"),
                                           ("dyn_masked_granular", "granular")])
def test_dynamics_masked_vs_reference_golden(ag, dev, name, material):
    g = load_golden(name)
    task = task_of(g)
    m = _model(ag, g, material, dev)
    out = ag.dynamics_masked(torch.from_numpy(g["state_init"]).to(dev), torch.from_numpy(g["state_mask"]).to(dev),
                             torch.from_numpy(g["action"]).to(dev), m, dev, _ppm(task, material))
    assert torch.equal(out["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    err = np.abs(out["state_seqs"].cpu().numpy() - g["state_seqs"]).max()
    assert err <= POS_TOL, err


def test_overflow_raises_like_reference(ag, dev):
    g = load_golden("dyn_overflow")
    task = task_of(g)
    m = _model(ag, g, "rope", dev)
    with pytest.raises(Exception, match="Exceeds max dims"):
        ag.dynamics(torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev), m, dev,
                    _ppm(task, "rope"))
    # the engine stays usable afterwards
    task["max_nR"] = 4000
    out = ag.dynamics(torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev), m, dev,
                      _ppm(task, "rope"))
    assert torch.isfinite(out["state_seqs"]).all()


def test_dynamics_vs_oracle_free_running_20_steps(ag, dev):
    """BASELINE north-star tolerance on a 20-step free-running rollout (2 look-ahead x repeat 10), granular-like cloud,
    oracle as the checker: <=1e-5 max-abs position error with bit-identical edge sets at every step."""
    from oracle import adaptigraph_oracle as O
    rng = np.random.default_rng(21)
    W = O.random_weights(21)
    task = dict(adj_thresh=0.4, topk=20, connect_tools_all=False, sim_real_ratio=10, push_length=0.2,
                gripper_enable=False, max_n=1, max_nR=20000, n_his=4, eef_num=5, material="granular",
                pusher_points=[[0, 0, 0.1], [0, 0.05, 0.1], [0, 0.025, 0.1], [0, -0.025, 0.1], [0, -0.05, 0.1]],
                material_dims={"granular": 1}, material_indices={"granular": 0})
    side = 18
    gx = (np.arange(side) - side / 2) * 0.12
    xx, zz = np.meshgrid(gx, gx, indexing="ij")
    cloud = (np.stack([xx.ravel() - 2, np.zeros(side * side), zz.ravel() + 1], 1) +
             rng.normal(0, 0.02, (side * side, 3))).astype(np.float32)
    B, H = 2, 2
    action = np.zeros((B, H, 4), np.float32)
    action[..., 0] = -2 + rng.uniform(-0.5, 0.5, (B, H))
    action[..., 1] = 1 + rng.uniform(-0.5, 0.5, (B, H))
    action[..., 2] = rng.uniform(-3, 3, (B, H))
    action[..., 3] = 10.5
    trace = []
    want = O.dynamics(W, 3, cloud, action, task, trace=trace)
    m = ag.DynamicsPredictor(*_cfg("granular", 3), dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(action).to(dev), m, dev, _ppm(task, "granular"))
    err = np.abs(out["state_seqs"].cpu().numpy() - want["state_seqs"]).max()
    assert err <= POS_TOL, err


def test_dynamics_repeat_zero_vs_reference_golden(ag, dev):
    """action_repeat == 0 (forward_dynamics.py:32,38) against the reference's own output."""
    g = load_golden("dyn_rope_repeat0")
    task = task_of(g)
    m = _model(ag, g, "rope", dev)
    out = ag.dynamics(torch.from_numpy(g["state0"]).to(dev), torch.from_numpy(g["action"]).to(dev), m, dev, _ppm(task, "rope"))
    got = out["state_seqs"].cpu().numpy()
    assert np.all(got[0, 0] == 0) and np.all(got[1, 1] == 0)
    assert torch.equal(out["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    assert np.abs(got - g["state_seqs"]).max() <= POS_TOL


def test_forward_softbody_variant_nhis5_vs_reference_golden(ag, dev):
    """config/dynamics/softbody.yaml: n_his = 5 (rel_input_dim 20), pstep = 4 - the eval-rollout path's model(**graph).
    The bf16x3 arithmetic refuses such a model loudly; the rollout driver serves it since r03
    (tests/test_gpu_more.py::test_rollout_with_the_softbody_model_variant_vs_reference_golden)."""
    g = load_golden("forward_softbody_nhis5")
    mc, mat, ds = _cfg("softbody", int(g["pstep"]))
    ds = dict(ds, n_his=5)
    m = ag.DynamicsPredictor(mc, mat, ds, dev)
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("w::")})
    edges = split_edges(g, "")
    B, N = g["attrs"].shape[:2]
    E = max(len(r) for r, _ in edges)
    Rr, Rs = torch.zeros((B, E, N)), torch.zeros((B, E, N))
    for b, (r, s) in enumerate(edges):
        Rr[b, torch.arange(len(r)), torch.from_numpy(r).long()] = 1
        Rs[b, torch.arange(len(s)), torch.from_numpy(s).long()] = 1
    kw = dict(state=torch.from_numpy(g["state"]).to(dev), attrs=torch.from_numpy(g["attrs"]).to(dev),
              p_instance=torch.from_numpy(g["p_instance"]).to(dev), action=torch.from_numpy(g["action"]).to(dev),
              softbody_physics_param=torch.from_numpy(g["physics_param"]).to(dev))
    pos, mot = m(Rr=Rr.to(dev), Rs=Rs.to(dev), **kw)
    assert np.abs(pos.cpu().numpy() - g["pred_pos"]).max() <= POS_TOL
    assert np.abs(mot.cpu().numpy() - g["pred_motion"]).max() <= POS_TOL
    with pytest.raises(NotImplementedError, match="n_his"):
        m.set_precision("bf16x3")
    m._precision = None
    task = dict(adj_thresh=0.4, topk=20, connect_tools_all=False, sim_real_ratio=10, push_length=0.2, gripper_enable=False,
                max_n=1, max_nR=9000, n_his=5, eef_num=1, material="softbody", pusher_points=[[0, 0, 0.1]],
                material_dims={"softbody": 1}, material_indices={"softbody": 0})
    cloud = torch.from_numpy(g["state"][0, -1, :40]).to(dev)
    out = ag.dynamics(cloud, torch.tensor([[[float(cloud[:, 0].mean()), float(cloud[:, 2].mean()), 0.3, 2.5]]], device=dev), m,
                      dev, _ppm(task, "softbody"))
    assert out["state_seqs"].shape == (1, 1, 40, 3) and torch.isfinite(out["state_seqs"]).all()
