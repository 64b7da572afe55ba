"""Shared first forward (-m gpu).  dynamics() broadcasts ONE start state to all candidates with a constant history
(reference src/planning/forward_dynamics.py:25) and then builds and encodes every candidate's graph separately (:125,
src/dynamics/gnn/model.py:249-303).  With option share_first the relation encoder runs once per call over the object-object
edges of the start state's tool-free graph and the first forward of every candidate reads those C rows.  A row's chain does
not depend on where it is computed, so the results must be IDENTICAL BITS to share_first = 0 - on every material, one and
several streams, host-decoded and device-planned actions, mixed repeats incl. 0 - and the edge counts must add up."""
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


LIMITS = dict(action_lower_lim=[-4.5, -2.5, -3.14, 0.0], action_upper_lim=[0.0, 4.5, 3.14, 7.0])


@pytest.mark.parametrize("material,cloud_fn,B,H", [
    ("rope", lambda r: _rope(200, r), 500, 1),                 # the shipped planner's chunk (planning/rope.yaml:39-42)
    ("granular", lambda r: _grid(14, 0.12, 0.02, r), 300, 1),  # 196 + 5 particles, top-k 20, five-point pusher
    ("cloth", lambda r: _grid(30, 0.3, 0.02, r), 150, 2),      # connect_tools_all: the tool sends to every particle
])
@pytest.mark.parametrize("device_plan", [False, True])
def test_shared_first_forward_is_bit_identical(ag, O, dev, material, cloud_fn, B, H, device_plan):
    rng = np.random.default_rng(211)
    task = _task(material, max_nR=40000, **(LIMITS if device_plan else {}))
    W, m = _model(ag, O, material, 211, dev)
    cloud = cloud_fn(rng)
    reps = rng.integers(1, 7, (B, H))
    reps[3, 0] = 0                                             # never live at the first forward
    a_np = _actions(cloud, B, H, reps, rng, spread=0.8)
    a_np[3, 0, 3] = 0.5
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, material)
    eng = m.engine(dev)
    outs = {}
    for streams in (1, 4):
        for chunk in (0, 67):
            with eng.options(streams=streams, device_decode=1 if device_plan else 0, share_prefix=0):
                eng.set_chunk(chunk)
                try:
                    with eng.options(share_first=1):
                        got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
                        base, shared, own = eng.share_counts()
                    with eng.options(share_first=0):
                        ref = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
                        assert eng.share_counts() == (0, 0, 0)
                finally:
                    eng.set_chunk(0)
            assert torch.isfinite(got).all() and torch.equal(got, ref), (streams, chunk)
            outs[(streams, chunk)] = (got, base, shared, own)
    first = outs[(1, 0)]
    # (a launch chunk small enough for the latency-mode propagate chains keeps its own C rows: that call reports no sharing)
    assert all(torch.equal(first[0], o[0]) and o[1:] in (first[1:], (0, 0, 0)) for o in outs.values())
    base, shared, own = first[1:]
    # every live candidate's object-object edges came from the base table: one encode instead of (B - 1) of them
    n_live = int((reps[:, 0] >= 1).sum())
    assert base > 0 and shared > 0.5 * n_live * base and shared <= n_live * base, (base, shared, own, n_live)
    tool_edges_max = (cloud.shape[0] + task["eef_num"]) * task["eef_num"] * 2 + task["eef_num"] ** 2
    assert own <= n_live * tool_edges_max, (own, n_live, tool_edges_max)
    print(f"{material}: first forward encodes {base} + {own} edges instead of {shared + own} "
          f"({(shared + own) / (base + own):.1f}x fewer), {n_live} live candidates")
    picks = [0, 3, B - 1]
    want = O.dynamics(W, 3, cloud, a_np[picks], task)["state_seqs"]
    err = np.abs(first[0][picks].cpu().numpy() - want).reshape(len(picks), -1).max(1)
    assert (err <= (POS_TOL if not device_plan else 5e-5)).sum() >= 2, err      # (a free-running rollout may pass a near-tie)
    assert float(first[0][3, 0].abs().max()) == 0.0


def test_shared_first_forward_auto_threshold_and_bf16x3(ag, O, dev):
    """share_first = -1 (default) shares from 8 candidates on and never for the latency-mode chains of a small launch; the
    bf16x3 arithmetic shares too (its chains are row-independent as well)."""
    rng = np.random.default_rng(223)
    task = _task("cloth", max_nR=40000)
    W, m = _model(ag, O, "cloth", 223, dev)
    cloud = _grid(30, 0.3, 0.02, rng)
    eng = m.engine(dev)
    ppm = _ppm(task, "cloth")
    s0 = torch.from_numpy(cloud).to(dev)
    for B, expect in ((4, False), (16, True), (40, True)):
        a = torch.from_numpy(_actions(cloud, B, 1, np.full(B, 2), rng))
        got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        base, shared, own = eng.share_counts()
        assert (base > 0) == expect and (shared > 0) == expect, (B, base, shared, own)
        with eng.options(share_first=0):
            assert torch.equal(got, ag.dynamics(s0, a, m, dev, ppm)["state_seqs"])
    with eng.options(latency=1, share_first=1):                 # latency-mode propagate chains: the step is not shared
        a = torch.from_numpy(_actions(cloud, 8, 1, np.full(8, 2), rng))
        got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        assert eng.share_counts() == (0, 0, 0)
    with eng.options(latency=0, share_first=0):
        assert torch.equal(got, ag.dynamics(s0, a, m, dev, ppm)["state_seqs"])
    m.set_precision("bf16x3")
    try:
        a = torch.from_numpy(_actions(cloud, 40, 2, rng.integers(1, 4, (40, 2)), rng))
        with eng.options(share_first=1):
            got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
            assert eng.share_counts()[1] > 0
        with eng.options(share_first=0):
            assert torch.equal(got, ag.dynamics(s0, a, m, dev, ppm)["state_seqs"])
    finally:
        m.set_precision("fp32")


def test_option_values_outside_their_range_are_refused(ag, O, dev):
    W, m = _model(ag, O, "rope", 5, dev)
    eng = m.engine(dev)
    for name, bad in (("stagger_us", -1), ("latency", 7), ("edge_block_min", -5), ("streams", 9), ("share_first", 2),
                      ("edge_wgs", 0)):
        before = eng.get_option(name)
        with pytest.raises(AssertionError, match="outside"):
            eng.set_option(name, bad)
        assert eng.get_option(name) == before


def test_device_planned_rollout_stops_enqueuing_at_the_chunk_maximum(ag, O, dev):
    """ag_rollout_actions enqueues a look-ahead step at most action_upper_lim[3] times; the chunk maxima come back to the host
    asynchronously (pinned memory + event, polled, never waited for) and the loop then stops at them: a 500-candidate call whose
    lengths are all <= 6 under a bound of 15 enqueues about 6 steps per chunk instead of 15.  Same bits as the host-planned path
    on the same decoded actions is not required here (device cos/sin); the results must equal the full-bound run bit for bit."""
    rng = np.random.default_rng(307)
    task = _task("rope", max_nR=40000, action_lower_lim=[-4.5, -2.5, -3.14, 2.0], action_upper_lim=[0.0, 4.5, 3.14, 15.0])
    W, m = _model(ag, O, "rope", 307, dev)
    cloud = _rope(200, rng)
    B = 500
    reps = rng.integers(2, 7, (B, 1))
    a = torch.from_numpy(_actions(cloud, B, 1, reps, rng, spread=0.8)).to(dev)
    s0 = torch.from_numpy(cloud).to(dev)
    ppm = _ppm(task, "rope")
    eng = m.engine(dev)
    for streams in (1, 4):
        with eng.options(streams=streams, share_prefix=0):
            torch.cuda.synchronize()
            got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
            enq, bound = eng.launch_counts()
            ex, need = eng.rollout_counts()
        n_chunks = bound // 15
        assert bound == 15 * n_chunks and need == int(reps.sum()) and ex == need
        # per chunk: its own maximum (6), plus whatever was enqueued before the plan's maxima landed (first chunk only)
        assert 6 * n_chunks <= enq <= 6 * n_chunks + 9, (enq, bound, n_chunks)
        print(f"streams {streams}: {enq} steps enqueued for {n_chunks} chunk(s) (bound alone: {bound})")
        tight = dict(task, action_upper_lim=[0.0, 4.5, 3.14, 6.0])
        ref = ag.dynamics(s0, a, m, dev, _ppm(tight, "rope"))["state_seqs"]
        assert torch.equal(got, ref)
