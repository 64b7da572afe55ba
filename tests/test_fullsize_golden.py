"""The oracle against the REAL reference at the headline sizes (CPU; fixtures: tests/golden/make_golden.py --fullsize).

full_cloth_a / full_cloth_flip: cloth 2025+1 particles, four candidates of bench.py's timed 1024-candidate batch (0, 1023
and the two whose GPU rollout left the oracle's after a near-tie in BENCH_r02: 49, 487), 20 free-running steps;
full_granular: granular 1024+5 particles (top-k 20), candidate 0 of tools/bench_configs.py's batch, 20 steps;
full_granular_b / _c (r05): candidates 85, 170 / 255, 128 of the same batch.

Bars: at EVERY forward the oracle's edge builder, fed the positions the reference fed its own, returns the reference's
edge list bit for bit; a single forward from the reference's own history is within 5e-6; free-running the oracle stays
within 1e-5 of the reference through all 20 steps - or, where it does not, the step at which it leaves is a near-tie in
the edge selection (margin < 4*thr*tol in squared distance): two correct fp32 implementations then follow different,
equally valid graphs.  That is what the reference ITSELF does against the oracle on the granular case (forward 18,
margin 1.0e-7), which puts the GPU path's three attributed flips of BENCH_r02 in context.
"""
import numpy as np
import pytest

from oracle import adaptigraph_oracle as O
from helpers import load_golden, task_of, fullsize_records

POS_TOL_FWD = 5e-6     # one forward, identical inputs (BLAS summation order: torch-MKL vs numpy)
POS_TOL = 1e-5         # BASELINE north star, free-running


def _masks(N_o, M, obj_mask=None):
    mask = np.ones(N_o + M, bool)
    if obj_mask is not None:
        mask[:N_o] = obj_mask
    tool = np.zeros(N_o + M, bool)
    tool[N_o:] = True
    return mask, tool


@pytest.mark.parametrize("name,expect", [
    ("full_cloth_a", {0: "ok", 1023: "ok"}),
    ("full_cloth_flip", {49: "ok", 487: "ok"}),      # the reference does NOT flip where the GPU did: the oracle follows it
    ("full_granular", {0: "tie@18"}),                # the reference and the oracle part at a 1e-7 near-tie
    # r05: four more candidates of tools/bench_configs.py's granular batch (BASELINE configs[2]).  Dense 20-nearest selection
    # among ~45 in-radius neighbours: over 20 forwards x 1029 receivers near-ties are the norm - the reference's own smallest
    # margins are 1.6e-7, 1.4e-7, 0 (an exact tie outside the k-th boundary) and 1.5e-8 - and the oracle stays on the reference's
    # graph through all 20 forwards in one candidate of the five (128)
    ("full_granular_b", {85: "tie@20", 170: "tie@18"}),
    ("full_granular_c", {255: "tie@15", 128: "ok"}),
    ("full_rope", {0: "ok", 21: "ok", 42: "ok", 63: "ok"}),          # BASELINE configs[1] size: rope 300+1, top-k 10 binding
    ("full_masked_cloth", {0: "ok", 1: "ok"}),       # dynamics_masked at size: 1400 and 2025 valid particles of 2025
])
def test_oracle_vs_reference_at_full_size(name, expect):
    g = load_golden(name)
    W, task = O.weights_from_npz(g), task_of(g)
    N_o, M = g["state0"].shape[0], task["eef_num"]
    masked = "state_mask" in g.files
    per_cand = fullsize_records(g, task)
    # 1. teacher-forced edges at every forward, bit-exact; selection margin of the reference's own positions
    margins = []
    for b, (recs, _, _) in enumerate(per_cand):
        mask, tool = _masks(N_o, M, g["state_mask"][b] if masked else None)
        mg = []
        for f, rec in enumerate(recs):
            r, s = O.construct_edges_single(rec["state_last"], task["adj_thresh"], mask, tool, task["topk"],
                                            task["connect_tools_all"])
            assert np.array_equal(r, rec["recv"]) and np.array_equal(s, rec["send"]), (name, b, f)
            mg.append(O.selection_margin(rec["state_last"], task["adj_thresh"], mask, tool, task["topk"]))
        margins.append(mg)
    # 2. free-running oracle
    tr = []
    if masked:
        out = O.dynamics_masked(W, int(g["pstep"]), g["state_init"], g["state_mask"], g["action"][:, 0], task, trace=tr)
        out = {k: v[:, None] for k, v in out.items()}
    else:
        out = O.dynamics(W, int(g["pstep"]), g["state0"], g["action"], task, trace=tr)
    assert np.array_equal(out["action_seqs"], g["action_seqs"])
    tie_margin = 4.0 * task["adj_thresh"] * POS_TOL
    for b, (recs, capture, want) in enumerate(per_cand):
        cand = int(g["cand_ids"][b])
        err = [float(np.abs(tr[b][f]["pred_pos"] - recs[f]["pred_pos"]).max()) for f in range(len(recs))]
        same = [np.array_equal(tr[b][f]["recv"], recs[f]["recv"]) and np.array_equal(tr[b][f]["send"], recs[f]["send"])
                for f in range(len(recs))]
        on_same = err
        if max(err) <= POS_TOL:
            verdict = "ok"
            assert all(same)
            for li, c in enumerate(capture):
                assert np.abs(out["state_seqs"][b, li] - want[li]).max() <= POS_TOL
        else:
            k = next(f for f, e in enumerate(err) if e > POS_TOL)
            assert k >= 1 and max(err[:k]) <= POS_TOL and all(same[:k]) and not same[k], (name, cand, k, err)
            # the graph the reference built at forward k hangs on a near-tie of its own distances
            assert margins[b][k] < tie_margin, (name, cand, k, margins[b][k])
            verdict, on_same = f"tie@{k + 1}", err[:k]
        print(f"{name} candidate {cand}: {verdict}; max error while on the same graph "
              f"{max(on_same):.2e}; smallest selection margin {min(margins[b]):.2e}")
        assert verdict == expect[cand], (name, cand, verdict, err)


@pytest.mark.parametrize("name", ["full_cloth_a", "full_granular"])
def test_oracle_single_forward_from_reference_history(name):
    """Forwards 1, 10 and 20 with the history the REFERENCE held (no free-running drift, no edge decision involved)."""
    g = load_golden(name)
    W, task = O.weights_from_npz(g), task_of(g)
    N_o, M = g["state0"].shape[0], task["eef_num"]
    N = N_o + M
    recs, capture, _ = fullsize_records(g, task)[0]
    first = [0] + [c + 1 for c in capture[:-1]]
    attrs = np.zeros((N, 2), np.float32)
    attrs[:N_o, 0] = 1
    attrs[N_o:, 1] = 1
    group = np.zeros((N, 1), np.float32)
    group[:N_o] = 1
    phys = np.zeros(N, np.float32)
    phys[:N_o] = 0.5
    dec, _ = O.decode_action(g["action"], task["push_length"])
    _, delta = O.tool_keypoints(dec, g["action"][..., 2], task)
    for f in (0, 9, 19):
        li = max(i for i, s in enumerate(first) if s <= f)
        hist = np.stack([recs[max(first[li], f - 3 + h)]["state_last"] for h in range(4)])     # forward_dynamics.py:83-85,176
        action = np.zeros((N, 3), np.float32)
        action[N_o:] = delta[0, li]
        pred, _ = O.model_forward_single(W, hist, attrs, recs[f]["recv"], recs[f]["send"], group, action, phys, int(g["pstep"]))
        err = float(np.abs(pred[:N_o] - recs[f]["pred_pos"]).max())
        assert err <= POS_TOL_FWD, (name, f, err)
