"""GPU parity at the sizes that are benchmarked (-m gpu): every BASELINE.json config at its own batch x horizon, and
the code path the headline number is measured on (two in-library streams, several launch chunks per stream,
second workspace carve, self-loop constant rows behind the last chunk) against the oracle.

The dense reference cannot run these sizes; the numpy oracle (pinned on the reference's goldens,
tests/test_oracle_vs_golden.py) rolls out a few candidates of each batch, and size-independent properties cover the
rest: candidates are independent, so any sub-batch, chunking or stream count must give the same bits per candidate.

Protocol for long free-running rollouts (SURVEY §7 "free-running drift vs. edge flips").  Over 20 steps on ~2k
particles there are tens of thousands of top-k / radius decisions; some are near-ties (two senders whose squared
distances differ by < 1e-6), and a position difference of one ulp between two correct implementations flips them,
after which the two rollouts follow different - equally valid - graphs.  So a candidate is checked like this:
  1. edges: at EVERY step the GPU builder, fed the oracle's positions, returns the oracle's edge list bit for bit;
  2. positions: free-running, max-abs error <= 1e-5 through all steps - or, if it leaves the tolerance at step k, then
     every earlier step is within tolerance AND the graphs of the two rollouts at step k differ only in pairs that are
     near-ties in the oracle's own distances (margin < 4*thr*tol: a position change within the tolerance flips them).
At least one candidate per case must pass (2) without any flip.
"""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import _ppm, POS_TOL
from test_gpu_more import _task, _grid, _rope, _actions, _model

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


def _chunk_plan(B, N, streams=2):
    """Mirror of ag_rollout_async's launch plan (csrc/ag_api.hip: auto_chunk + equal-chunk re-division) for the default
    settings - used only to pick WHICH candidates to hand to the oracle (first and last of every chunk)."""
    bc = max(1, min(B, (4 * 256 * 256) // N))
    ns = streams if B * N >= 65536 else 1
    if ns > 1:
        bc = min(bc, (B + ns - 1) // ns)
    n_chunks = (B + bc - 1) // bc
    if ns > 1:
        n_chunks = (n_chunks + ns - 1) // ns * ns
    bc = (B + n_chunks - 1) // n_chunks
    return [(b0, min(B, b0 + bc)) for b0 in range(0, B, bc)], ns


# ------------------------------------------------------------------------------------------------- the protocol
def _edge_set(el):
    n = int(el.n_edges[0])
    return el.recv[0, :n].cpu().numpy(), el.send[0, :n].cpu().numpy()


def _check_candidate(ag, O, dev, task, got, trace, per_step_fn, N_o, obj_mask=None, label=""):
    """got: GPU result of this candidate, list of (N_o,3) arrays per look-ahead step; trace: the oracle's per-forward
    records of the same candidate with `capture` = indices of the forwards whose prediction is captured;
    per_step_fn() -> (n_forwards, N_o, 3) GPU states after every forward (computed only when needed).
    Returns 'ok' or 'tie@k'; asserts on anything unexplained."""
    recs, capture, want = trace
    thr, topk, cta = task["adj_thresh"], task["topk"], task["connect_tools_all"]
    N = recs[0]["state_last"].shape[0]
    mask = np.ones(N, bool)
    if obj_mask is not None:
        mask[:N_o] = obj_mask
    tool = np.zeros(N, bool)
    tool[N_o:] = True
    mask_t, tool_t = torch.from_numpy(mask[None]).to(dev), torch.from_numpy(tool[None]).to(dev)
    # 1. teacher-forced edges, every step, bit-exact
    for k, rec in enumerate(recs):
        el = ag.construct_edges_index(torch.from_numpy(rec["state_last"][None]).to(dev), thr, mask_t, tool_t, topk, cta)
        r, s = _edge_set(el)
        assert np.array_equal(r, rec["recv"]) and np.array_equal(s, rec["send"]), f"{label}: edges differ at step {k + 1}"
    # 2. free-running positions
    errs = [float(np.abs(g - w).max()) for g, w in zip(got, want)]
    if max(errs) <= POS_TOL:
        return "ok", max(errs)
    steps = per_step_fn()
    assert len(steps) == len(recs)
    for li, c in enumerate(capture):                                    # the per-step states ARE the rollout's states
        assert np.array_equal(steps[c], got[li]), f"{label}: per-step replay differs from the rollout at capture {li}"
    e = [float(np.abs(steps[k] - recs[k]["pred_pos"]).max()) for k in range(len(recs))]
    k = next(i for i, v in enumerate(e) if v > POS_TOL)
    assert k >= 1, f"{label}: error {e[k]:.2e} at the very first forward (identical inputs: cannot be an edge flip)"
    assert max(e[:k]) <= POS_TOL
    # graphs at forward k: the oracle's, and the GPU builder's on the GPU rollout's own positions
    pos_g = recs[k]["state_last"].copy()
    pos_g[:N_o] = steps[k - 1]
    if obj_mask is None:                                                # tool height follows the GPU's own cloud (:163)
        y = np.float32(steps[k - 1][:, 1].min())
        if task["gripper_enable"]:
            y = np.float32(y + np.float32(0.01 * task["sim_real_ratio"]))
        pos_g[N_o:, 1] = y
    el = ag.construct_edges_index(torch.from_numpy(pos_g[None]).to(dev), thr, mask_t, tool_t, topk, cta)
    r, s = _edge_set(el)
    eo = set(zip(recs[k]["recv"].tolist(), recs[k]["send"].tolist()))
    eg = set(zip(r.tolist(), s.tolist()))
    diff = eo ^ eg
    assert diff, f"{label}: error {e[k]:.2e} at step {k + 1} with identical graphs - not an edge flip"
    margin = 4.0 * thr * POS_TOL
    dis = O.pairwise_dis(recs[k]["state_last"])
    thr2 = np.float32(np.float32(thr) * np.float32(thr))
    by_row = {}
    for (i, j) in diff:
        by_row.setdefault(i, []).append(j)
    for i, js in by_row.items():
        for j in js:
            near_radius = abs(float(dis[i, j]) - float(thr2)) < margin
            near_swap = any(abs(float(dis[i, j]) - float(dis[i, j2])) < margin for j2 in js if j2 != j)
            tool_rule = cta and (tool[i] or tool[j])                    # the all-or-nothing tool rule hangs on ONE radius test
            if tool_rule:
                t_rows = dis[np.ix_(tool, mask & ~tool)]
                tool_rule = float(np.abs(t_rows - thr2).min()) < margin
            assert near_radius or near_swap or tool_rule, \
                f"{label}: step {k + 1} pair ({i},{j}) flipped without being a near-tie (dis {dis[i, j]:.9f})"
    return f"tie@{k + 1}", max(e[:k])


def _oracle_trace(O, W, task, cloud, act, masked=None):
    """One candidate through the oracle with a trace; returns (records, capture indices, want list)."""
    tr = []
    if masked is None:
        want = O.dynamics(W, 3, cloud, act[None], task, trace=tr)["state_seqs"][0]
        _, rep = O.decode_action(act[None], task["push_length"])
        capture = (np.cumsum(rep[0]) - 1).tolist()
        return (tr[0], capture, list(want))
    state, mask = masked
    want = O.dynamics_masked(W, 3, state[None], mask[None], act[None], task, trace=tr)["state_seqs"][0]
    return (tr[0], [len(tr[0]) - 1], [want])


def _per_step_unmasked(ag, m, dev, ppm, cloud, act):
    """Same push, repeat = 1..R: candidate r holds the state after r forwards of that look-ahead step."""
    s0 = torch.from_numpy(cloud).to(dev)
    out = []
    for li in range(act.shape[0]):
        R = int(act[li, 3])
        a = np.repeat(act[None, :li + 1], R, 0).copy()
        a[:, li, 3] = np.arange(1, R + 1) + 0.5
        out.append(ag.dynamics(s0, torch.from_numpy(a).to(dev), m, dev, ppm)["state_seqs"][:, li].cpu().numpy())
    return np.concatenate(out, 0)


def _per_step_masked(ag, m, dev, ppm, state, mask, act):
    R = int(act[3])
    a = np.repeat(act[None], R, 0).copy()
    a[:, 3] = np.arange(1, R + 1) + 0.5
    st = torch.from_numpy(np.repeat(state[None], R, 0)).to(dev)
    mk = torch.from_numpy(np.repeat(mask[None], R, 0)).to(dev)
    return ag.dynamics_masked(st, mk, torch.from_numpy(a).to(dev), m, dev, ppm)["state_seqs"].cpu().numpy()


def _check_picks(ag, O, dev, W, m, task, material, picks, seq, fn_trace, fn_steps, N_o, fn_mask=None, min_clean=1):
    verdicts = []
    for b in picks:
        v, err = _check_candidate(ag, O, dev, task, [x for x in seq[b]], fn_trace(b), lambda b=b: fn_steps(b), N_o,
                                  obj_mask=fn_mask(b) if fn_mask else None, label=f"{material} candidate {b}")
        verdicts.append((b, v, err))
    print(f"{material}: " + ", ".join(f"cand {b}: {v} (err within tolerance part {e:.2e})" for b, v, e in verdicts))
    assert sum(v == "ok" for _, v, _ in verdicts) >= min_clean, verdicts
    return verdicts


# ------------------------------------------------------------------------------------------------- the tests
def test_two_stream_multi_chunk_path_vs_oracle_and_one_stream(ag, O, dev):
    """The benchmarked path: cloth 2025+1 particles, enough candidates for two streams x two chunks each (default AG_*).
    First and last candidate of EVERY chunk against the oracle; the whole tensor bit-for-bit against the same call on
    one stream, against one-stream small chunks, and against small-batch calls (the path the goldens pin)."""
    rng = np.random.default_rng(41)
    task = _task("cloth")
    W, m = _model(ag, O, "cloth", 41, dev)
    cloud = _grid(45, 0.3, 0.02, rng)
    N = cloud.shape[0] + 1
    B, H, rep = 320, 2, 3
    chunks, ns = _chunk_plan(B, N)
    assert ns == 2 and len(chunks) >= 4, (chunks, ns)                   # two streams, >= 2 chunks per stream
    a_np = _actions(cloud, B, H, rep, rng, spread=2.0)
    a_np[::7, 0, 3] = 2.5                                               # mixed repeats inside every chunk
    a_np[3::11, 1, 3] = 1.5
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, "cloth")
    eng = m.engine(dev)
    eng.set_chunk(0)
    with eng.options(streams=2):
        two = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    assert torch.isfinite(two).all()
    with eng.options(streams=1):
        one = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        eng.set_chunk(37)                                               # odd chunk size, short last chunk, one stream
        odd = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        eng.set_chunk(0)
    assert torch.equal(two, one)
    assert torch.equal(two, odd)
    picks = sorted({b for lo, hi in chunks for b in (lo, hi - 1)})
    small = ag.dynamics(s0, a[picks], m, dev, ppm)["state_seqs"]        # 8 candidates: one stream, one chunk
    assert torch.equal(two[picks], small)
    seq = two.cpu().numpy()
    _check_picks(ag, O, dev, W, m, task, "cloth two-stream", picks, seq,
                 lambda b: _oracle_trace(O, W, task, cloud, a_np[b]),
                 lambda b: _per_step_unmasked(ag, m, dev, ppm, cloud, a_np[b]), cloud.shape[0], min_clean=len(picks) // 2)


def _property_checks(ag, m, dev, s0, a, ppm, seq):
    """Size-independent: a sub-batch that straddles a chunk boundary and a different chunking give the same bits."""
    B = a.shape[0]
    lo = max(0, B // 2 - 3)
    part = ag.dynamics(s0, a[lo:lo + 7], m, dev, ppm)["state_seqs"]
    assert torch.equal(part, seq[lo:lo + 7])
    eng = m.engine(dev)
    eng.set_chunk(max(1, B // 5 + 1))
    try:
        again = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    finally:
        eng.set_chunk(0)
    assert torch.equal(again, seq)


@pytest.mark.parametrize("material,cloud_fn,B", [
    ("rope", lambda r: _rope(300, r), 64),                              # BASELINE configs[1]
    ("granular", lambda r: _grid(32, 0.12, 0.02, r), 256),              # configs[2]: 1024+5 particles, top-k 20
    ("cloth", lambda r: _grid(45, 0.3, 0.02, r), 1024),                 # configs[3]: the bench line's batch
])
def test_baseline_configs_at_their_own_size(ag, O, dev, material, cloud_fn, B):
    """B x 20 rollout steps (2 look-ahead x repeat 10) exactly as bench.py / tools/bench_configs.py time them."""
    rng = np.random.default_rng(43)
    task = _task(material, max_nR=40000)
    W, m = _model(ag, O, material, 43, dev)
    cloud = cloud_fn(rng)
    a_np = _actions(cloud, B, 2, 10, rng, spread=1.5 if material != "rope" else 0.6)
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, material)
    seq_t = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    assert seq_t.shape == (B, 2, cloud.shape[0], 3) and torch.isfinite(seq_t).all()
    _property_checks(ag, m, dev, s0, a, ppm, seq_t)
    _check_picks(ag, O, dev, W, m, task, f"{material} {B} x 20", [0, B - 1, B // 2], seq_t.cpu().numpy(),
                 lambda b: _oracle_trace(O, W, task, cloud, a_np[b]),
                 lambda b: _per_step_unmasked(ag, m, dev, ppm, cloud, a_np[b]), cloud.shape[0])


@pytest.mark.parametrize("material,cloud_fn,B", [
    ("rope", lambda r: _rope(300, r), 172),
    ("granular", lambda r: _grid(32, 0.12, 0.02, r), 170),
    ("cloth", lambda r: _grid(45, 0.3, 0.02, r), 170),
])
def test_config4_mixed_variable_size_batch(ag, O, dev, material, cloud_fn, B):
    """BASELINE configs[4]: 512 candidates = 172 rope + 170 granular + 170 cloth, every candidate with
    its own particle count N_o ~ U{0.5 N .. N} (dynamics_masked-style padding, one model context per material),
    20 rollout steps.  Oracle on the smallest, the largest and a middle-sized candidate of each material."""
    rng = np.random.default_rng(47)
    task = _task(material, max_nR=40000)
    W, m = _model(ag, O, material, 47, dev)
    cloud = cloud_fn(rng)
    N = cloud.shape[0]
    counts = rng.integers(N // 2, N + 1, B)
    counts[0], counts[-1] = N, N // 2                                   # both extremes are present
    state = np.zeros((B, N, 3), np.float32)
    mask = np.zeros((B, N), bool)
    for b, c in enumerate(counts):
        keep = np.sort(rng.choice(N, c, replace=False))                 # a random subset, moved to the front (prefix mask)
        state[b, :c] = cloud[keep]
        mask[b, :c] = True
    a_np = _actions(cloud, B, 1, 20, rng, spread=1.5 if material != "rope" else 0.6)[:, 0]
    a_np[1::5, 3] = 12.5                                                # mixed repeats: some candidates stop earlier
    args = (torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a_np).to(dev))
    ppm = _ppm(task, material)
    seq = ag.dynamics_masked(*args, m, dev, ppm)["state_seqs"]
    assert seq.shape == (B, N, 3) and torch.isfinite(seq).all()
    order = np.argsort(counts, kind="stable")
    picks = sorted({int(order[0]), int(order[-1]), int(order[B // 2])})
    sub = ag.dynamics_masked(*(t[picks] for t in args), m, dev, ppm)["state_seqs"]
    assert torch.equal(sub, seq[picks])                                 # independent of the rest of the batch
    print(f"cfg4 {material}: particle counts of the checked candidates {counts[picks].tolist()}")
    _check_picks(ag, O, dev, W, m, task, f"cfg4 {material}", picks, seq.cpu().numpy()[:, None],
                 lambda b: _oracle_trace(O, W, task, None, a_np[b], masked=(state[b], mask[b])),
                 lambda b: _per_step_masked(ag, m, dev, ppm, state[b], mask[b], a_np[b]), N,
                 fn_mask=lambda b: mask[b])


def test_config4_as_one_mixed_call_equals_the_three_calls_bitwise(ag, O, dev):
    """adaptigraph_amd.dynamics_mixed: the 172 rope + 170 granular + 170 cloth variable-size graphs of BASELINE configs[4] in ONE
    entry (the three materials dealt to three streams, one read-back of all flags) - every material's rows equal the stand-alone
    dynamics_masked call's (which the test above checks against the oracle), whatever runs beside them; an overflowing graph in
    ONE material raises the reference's Exception("Exceeds max dims") out of the one entry."""
    rng = np.random.default_rng(47)
    batches, tight = [], []
    for material, cloud_fn, B in (("rope", lambda r: _rope(300, r), 172), ("granular", lambda r: _grid(32, 0.12, 0.02, r), 170),
                                  ("cloth", lambda r: _grid(45, 0.3, 0.02, r), 170)):
        task = _task(material, max_nR=40000)
        W, m = _model(ag, O, material, 47, dev)
        cloud = cloud_fn(rng)
        N = cloud.shape[0]
        counts = rng.integers(N // 2, N + 1, B)
        state = np.zeros((B, N, 3), np.float32)
        mask = np.zeros((B, N), bool)
        for b, c in enumerate(counts):
            state[b, :c] = cloud[np.sort(rng.choice(N, c, replace=False))]
            mask[b, :c] = True
        a_np = _actions(cloud, B, 1, 20, rng, spread=1.5 if material != "rope" else 0.6)[:, 0]
        a_np[1::5, 3] = 12.5
        batches.append((torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a_np), m, _ppm(task, material)))
        small = _task(material, max_nR=40000 if material != "granular" else 500)           # granular's graphs have ~20,000 edges
        tight.append(batches[-1][:4] + (_ppm(small, material),))
    want = [ag.dynamics_masked(b[0], b[1], b[2], b[3], dev, b[4])["state_seqs"] for b in batches]
    for pin in (None, True, False):
        got = ag.dynamics_mixed(batches, dev, one_stream_each=pin)
        assert len(got) == 3
        for w, g_, b in zip(want, got, batches):
            assert g_["state_seqs"].shape == w.shape and torch.equal(g_["state_seqs"], w), pin
            assert g_["action_seqs"].shape == (b[0].shape[0], 4)
    with pytest.raises(Exception, match="Exceeds max dims"):
        ag.dynamics_mixed(tight, dev)
    again = ag.dynamics_mixed(batches, dev)                              # the contexts are as good as before
    assert all(torch.equal(g_["state_seqs"], w) for g_, w in zip(again, want))
    assert ag.dynamics_mixed([], dev) == []
