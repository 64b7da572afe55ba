"""The HIP path against the REAL reference at the headline sizes (-m gpu; fixtures: tests/golden/make_golden.py
--fullsize: cloth 2025+1, four candidates of bench.py's timed batch, and granular 1024+5, 20 free-running steps each).

Protocol of tests/test_gpu_fullsize.py (_check_candidate) with the reference's records in the oracle's place:
  1. edges: at EVERY forward the GPU builder, fed the positions the reference fed its own, returns the reference's edge
     list bit for bit;
  2. positions: free-running within 1e-5 of the reference through all 20 steps, or - where the error leaves the tolerance at
     forward k - every earlier forward is within tolerance and the two graphs at forward k differ only in pairs that are
     near-ties in the reference's own distances;
  3. one forward from the reference's own history (forwards 1, 10, 20): within 1e-5, no edge decision involved.
Candidates 49 and 487 are the ones BENCH_r02 reported as attributed flips against the oracle; the reference does not flip
there (tests/test_fullsize_golden.py), the granular case is where the reference itself parts from the oracle.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, task_of, fullsize_records
from test_gpu_parity import _model, _ppm, POS_TOL
from test_gpu_fullsize import _check_candidate, _per_step_unmasked, _per_step_masked

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


# min_clean = what the runs so far produced (r03 logs gpurun_out/r3a, r3k; unchanged since: the later changes are bit-identical):
# every cloth candidate clean on both action paths, 3 of 4 rope candidates (candidate 0 parts at forward 10, an attributed
# tie - the rope's smallest selection margin is 1.9e-8), and 0 only for full_granular, where the REFERENCE itself parts
# from the oracle at forward 18 at a 1e-7 near-tie and the GPU follows the oracle's side.
# r05 (gpurun_out/r5d): four more candidates of BASELINE configs[2]'s batch, reference records from make_golden.py --fullsize-r05:
# the GPU stays within 2.9e-6 of the REFERENCE through all 20 forwards on three of them (85, 255, 128 - on both action paths; the
# numpy oracle stays with the reference on 128 only) and parts at forward 18 on candidate 170, a near-tie in the reference's own
# distances (its smallest selection margin there is 1.4e-7): min_clean 1 + 2
@pytest.mark.parametrize("name,material,min_clean", [("full_cloth_a", "cloth", 2), ("full_cloth_flip", "cloth", 2),
                                                     ("full_granular", "granular", 0), ("full_granular_b", "granular", 1),
                                                     ("full_granular_c", "granular", 2), ("full_rope", "rope", 3),
                                                     ("full_masked_cloth", "cloth", 2)])
def test_rollout_vs_reference_at_full_size(ag, O, dev, name, material, min_clean):
    g = load_golden(name)
    task = task_of(g)
    m = _model(ag, g, material, dev)
    ppm = _ppm(task, material)
    cloud, act = g["state0"], g["action"]
    masked = "state_mask" in g.files                                    # dynamics_masked: per-candidate clouds + masks
    if masked:
        st, mk = g["state_init"], g["state_mask"]
        out = ag.dynamics_masked(torch.from_numpy(st).to(dev), torch.from_numpy(mk).to(dev), torch.from_numpy(act[:, 0]).to(dev),
                                 m, dev, ppm)
        out = {k: v[:, None] for k, v in out.items()}
    else:
        out = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(act).to(dev), m, dev, ppm)
    assert torch.equal(out["action_seqs"].cpu(), torch.from_numpy(g["action_seqs"]))
    seq = out["state_seqs"].cpu().numpy()
    verdicts = []
    for b, trace in enumerate(fullsize_records(g, task)):
        steps_fn = (lambda b=b: _per_step_masked(ag, m, dev, ppm, st[b], mk[b], act[b, 0])) if masked else \
                   (lambda b=b: _per_step_unmasked(ag, m, dev, ppm, cloud, act[b]))
        v, err = _check_candidate(ag, O, dev, task, [x for x in seq[b]], trace, steps_fn, cloud.shape[0],
                                  obj_mask=mk[b] if masked else None,
                                  label=f"{name} candidate {int(g['cand_ids'][b])}")
        verdicts.append((int(g["cand_ids"][b]), v, err))
    print(f"{name} vs the reference: " + ", ".join(f"cand {c}: {v} (err while within tolerance {e:.2e})" for c, v, e in verdicts))
    assert sum(v == "ok" for _, v, _ in verdicts) >= min_clean, verdicts
    if masked:
        return
    # the device-planned action path (ag_rollout_actions: decode, tool layout - 1-point / 5-point pusher, gripper - and launch
    # plan on the GPU) against the same reference records: decoded actions to 1e-6, states by the same protocol
    tdev = dict(task, action_upper_lim=[0.0, 4.5, 3.14, float(int(act[..., 3].max()))])
    pdev = _ppm(tdev, material)
    out_d = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(act).to(dev), m, dev, pdev)
    assert float((out_d["action_seqs"].cpu() - torch.from_numpy(g["action_seqs"])).abs().max()) <= 1e-6
    seq_d = out_d["state_seqs"].cpu().numpy()
    vd = []
    for b, trace in enumerate(fullsize_records(g, task)):
        v, err = _check_candidate(ag, O, dev, tdev, [x for x in seq_d[b]], trace,
                                  lambda b=b: _per_step_unmasked(ag, m, dev, pdev, cloud, act[b]), cloud.shape[0],
                                  label=f"{name} candidate {int(g['cand_ids'][b])} (device-planned)")
        vd.append((int(g["cand_ids"][b]), v, err))
    print(f"{name} vs the reference, device-planned: " + ", ".join(f"cand {c}: {v} ({e:.2e})" for c, v, e in vd))
    assert sum(v == "ok" for _, v, _ in vd) >= min_clean, vd


@pytest.mark.parametrize("name,material", [("full_cloth_a", "cloth"), ("full_granular", "granular")])
def test_single_forward_from_reference_history(ag, O, dev, name, material):
    g = load_golden(name)
    task = task_of(g)
    m = _model(ag, g, material, dev)
    N_o, M = g["state0"].shape[0], task["eef_num"]
    N = N_o + M
    recs, capture, _ = fullsize_records(g, task)[0]
    first = [0] + [c + 1 for c in capture[:-1]]
    attrs = np.zeros((1, N, 2), np.float32)
    attrs[:, :N_o, 0] = 1
    attrs[:, N_o:, 1] = 1
    dec, _ = O.decode_action(g["action"], task["push_length"])
    _, delta = O.tool_keypoints(dec, g["action"][..., 2], task)
    mask = torch.ones((1, N), dtype=torch.bool, device=dev)
    tool = torch.zeros((1, N), dtype=torch.bool, device=dev)
    tool[:, N_o:] = True
    for f in (0, 9, 19):
        li = max(i for i, s in enumerate(first) if s <= f)
        hist = np.stack([recs[max(first[li], f - 3 + h)]["state_last"] for h in range(4)])[None]
        action = np.zeros((1, N, 3), np.float32)
        action[0, N_o:] = delta[0, li]
        state = torch.from_numpy(hist).to(dev)
        el = ag.construct_edges_index(state[:, -1], task["adj_thresh"], mask, tool, task["topk"], task["connect_tools_all"])
        n = int(el.n_edges[0])
        assert np.array_equal(el.recv[0, :n].cpu().numpy(), recs[f]["recv"]) and np.array_equal(el.send[0, :n].cpu().numpy(), recs[f]["send"])
        pos, _ = m(state=state, attrs=torch.from_numpy(attrs).to(dev), edges=el,
                   p_instance=torch.ones((1, N_o, 1), device=dev), action=torch.from_numpy(action).to(dev),
                   **{f"{material}_physics_param": torch.full((1, 1), 0.5, device=dev)})
        err = float(np.abs(pos[0].cpu().numpy() - recs[f]["pred_pos"]).max())
        assert err <= POS_TOL, (name, f, err)
