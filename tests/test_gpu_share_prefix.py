"""Contact-free prefix (-m gpu).  A tool acts on the object only through the edges it takes part in (reference
src/dynamics/dataset/graph.py:233-298: radius test, then top-k; connect_tools_all's tool -> object edges are all-or-nothing on
"some object sits inside a tool particle's radius", :276-286), and it takes part in none while no object particle is inside its radius.  Until then a candidate's object particles evolve exactly like the start state without a tool - bit for bit on
this engine, where a row's result does not depend on the rest of the batch.  With option share_prefix that base rollout runs
once per dynamics() call, every candidate is stepped only from its first contact on, and one that never touches takes the base
state of its last step (reference src/planning/forward_dynamics.py:156-176 steps all of them).  Everything must be IDENTICAL BITS
to share_prefix = 0: both pushers, both action paths, one and several streams, one and two look-ahead steps, repeats incl. 0."""
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


LIMITS = dict(action_lower_lim=[-4.5, -2.5, -3.14, 0.0], action_upper_lim=[0.0, 4.5, 3.14, 9.0])


def _first_contact(O, W, cloud, a, task):
    """first forward (1-based) of look-ahead step 0 whose graph holds a tool edge in the ORACLE's rollout, 0 = never"""
    tr = []
    O.dynamics(W, 3, cloud, a[None], task, trace=tr)
    N_o = cloud.shape[0]
    for f, rec in enumerate(tr[0][:int(a[0, 3])]):
        if ((rec["recv"] >= N_o) | (rec["send"] >= N_o)).any():
            return f + 1
    return 0


@pytest.mark.parametrize("material,cloud_fn,B,H,spread", [
    ("rope", lambda r: _rope(200, r), 500, 1, 2.5),            # the shipped planner's chunk; most pushes never reach the rope
    ("rope", lambda r: _rope(200, r), 300, 2, 1.2),            # two look-ahead steps: only the first shares
    ("granular", lambda r: _grid(14, 0.12, 0.02, r), 300, 1, 1.5),   # five-point pusher, top-k 20
    ("cloth", lambda r: _grid(16, 0.3, 0.02, r), 200, 1, 6.0),       # connect_tools_all: all-or-nothing on the same contact (graph.py:276-286)
])
@pytest.mark.parametrize("device_plan", [False, True])
def test_contact_free_prefix_is_bit_identical(ag, O, dev, material, cloud_fn, B, H, spread, device_plan):
    rng = np.random.default_rng(401)
    task = _task(material, max_nR=40000, **(LIMITS if device_plan else {}))
    W, m = _model(ag, O, material, 401, dev)
    cloud = cloud_fn(rng)
    reps = rng.integers(1, 9, (B, H))
    reps[3, 0] = 0
    a_np = _actions(cloud, B, H, reps, rng, spread=spread)
    a_np[3, 0, 3] = 0.5
    a_np[5, 0, :2] = cloud[100, [0, 2]]                         # starts on the object: contact at the first forward
    a_np[5, 0, 3] = max(a_np[5, 0, 3], 1.5)
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, material)
    eng = m.engine(dev)
    outs = {}
    for streams in (1, 4):
        for chunk in (0, 67):
            with eng.options(streams=streams, device_decode=1 if device_plan else 0):
                eng.set_chunk(chunk)
                try:
                    with eng.options(share_prefix=1):
                        got = ag.dynamics(s0, a, m, dev, ppm)
                        ex, need = eng.rollout_counts()
                    with eng.options(share_prefix=0):
                        ref = ag.dynamics(s0, a, m, dev, ppm)
                        ex0, need0 = eng.rollout_counts()
                finally:
                    eng.set_chunk(0)
            assert torch.isfinite(got["state_seqs"]).all()
            assert torch.equal(got["state_seqs"], ref["state_seqs"]) and torch.equal(got["action_seqs"], ref["action_seqs"]), (streams, chunk)
            assert need == need0 == int(reps.sum()) and ex0 == need0 and ex < need, (ex, need, ex0, need0)
            outs[(streams, chunk)] = (got["state_seqs"], ex)
    first = outs[(1, 0)]
    assert all(torch.equal(first[0], o[0]) and o[1] == first[1] for o in outs.values())
    print(f"{material} {B} x {H}: {first[1]} candidate-forwards executed (base rollout included) of {int(reps.sum())} the reference steps "
          f"({int(reps.sum()) / first[1]:.1f}x fewer)")
    assert float(first[0][3, 0].abs().max()) == 0.0            # repeat 0 at look-ahead step 0: zeros (:32)
    if device_plan:
        return
    # against the oracle: a candidate that touches at once, and the first few others - incl. where each one's first contact is
    picks = [5, 0, 1, 2, B - 1]
    want = O.dynamics(W, 3, cloud, a_np[picks], task)["state_seqs"]
    err = np.abs(first[0][picks].cpu().numpy() - want).reshape(len(picks), -1).max(1)
    assert (err <= POS_TOL).sum() >= len(picks) - 1, err        # (a free-running rollout may pass a near-tie)
    assert _first_contact(O, W, cloud, a_np[5], task) == 1
    contacts = [_first_contact(O, W, cloud, a_np[b], task) for b in range(12)]
    assert any(c == 0 for c in contacts), contacts              # the batch does hold candidates that never touch


def test_over_bound_repeats_are_treated_alike_with_and_without_prefix_sharing(ag, O, dev):
    """Device-planned calls take the caller's bound of action_repeat (task_config['action_upper_lim'][3]).  A candidate beyond it -
    also a garbage length like inf or 1e9, at look-ahead step 0 or later - is stepped at most `bound` times and never captured,
    the asynchronous shim marks its rows NaN and flags[1] reports it; the launch loop stays bounded.  The same with the contact-free
    prefix as without it (r04's prefix path copied later steps' repeats unclamped into its launch loop), and every other candidate
    is untouched by its neighbour's garbage."""
    rng = np.random.default_rng(467)
    task = _task("rope", max_nR=40000, **LIMITS)                  # bound 9
    W, m = _model(ag, O, "rope", 467, dev)
    cloud = _rope(200, rng)
    B, H = 200, 2
    reps = rng.integers(1, 9, (B, H))
    a_np = _actions(cloud, B, H, reps, rng, spread=2.0)
    a_np[7, 0, 3] = 12.5                                          # beyond the bound at look-ahead step 0
    a_np[9, 0, 3] = np.inf                                        # garbage
    a_np[11, 1, 3] = 1.0e9                                        # beyond the bound at look-ahead step 1
    bad = [7, 9, 11]
    good = [b for b in range(B) if b not in bad]
    s0, a = torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np).to(dev)
    ppm = _ppm(task, "rope")
    eng = m.engine(dev)
    outs = {}
    for sp in (1, 0):
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        with eng.options(share_prefix=sp, device_decode=1):
            out = ag.dynamics(s0, a, m, dev, ppm, _sync=False, _overflow_flag=flags)["state_seqs"]
            torch.cuda.synchronize()
            enq, bound = eng.launch_counts()
        assert flags[1].item() > 9 and flags[0].item() == 0
        assert enq <= bound <= 2 * 9 * 4, (enq, bound)             # no launch loop beyond bound x look-ahead steps x chunks
        assert torch.isnan(out[bad]).all() and torch.isfinite(out[good]).all()
        outs[sp] = out
    assert torch.equal(outs[1][good], outs[0][good])
    clean = torch.from_numpy(a_np[good]).to(dev)
    with eng.options(share_prefix=0, device_decode=1):
        assert torch.equal(ag.dynamics(s0, clean, m, dev, ppm)["state_seqs"], outs[0][good])


def test_prefix_sharing_when_every_or_no_candidate_touches(ag, O, dev):
    rng = np.random.default_rng(409)
    task = _task("rope", max_nR=40000)
    W, m = _model(ag, O, "rope", 409, dev)
    cloud = _rope(200, rng)
    B = 128
    eng = m.engine(dev)
    ppm = _ppm(task, "rope")
    s0 = torch.from_numpy(cloud).to(dev)
    reps = rng.integers(2, 6, (B, 1))
    for label, a_np in (("all", _actions(cloud, B, 1, reps, rng, spread=0.0)), ("none", _actions(cloud, B, 1, reps, rng, spread=0.0))):
        if label == "all":
            a_np[:, 0, :2] = cloud[rng.integers(0, 200, B)][:, [0, 2]]
        else:
            a_np[:, 0, 0] += 40.0
        a = torch.from_numpy(a_np)
        with eng.options(share_prefix=1):
            got = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
            ex, need = eng.rollout_counts()
        with eng.options(share_prefix=0):
            ref = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        assert torch.equal(got, ref), label
        base = int(reps.max())
        assert ex == (need + base if label == "all" else base), (label, ex, need, base)
        if label == "none":                                      # every candidate = the base rollout at its own repeat count
            by_rep = {int(r): got[i] for i, r in enumerate(reps[:, 0])}
            assert all(torch.equal(got[i], by_rep[int(r)]) for i, r in enumerate(reps[:, 0]))


def test_prefix_sharing_defaults(ag, O, dev):
    """auto (-1): from 64 candidates and 32768 rows on (and only while at most half of the candidates touch at the first forward)"""
    rng = np.random.default_rng(419)
    for material, cloud, B, expect in (("rope", _rope(200, rng), 32, False), ("rope", _rope(600, rng), 64, True),
                                       ("cloth", _grid(24, 0.3, 0.02, rng), 64, True)):
        task = _task(material, max_nR=60000)
        W, m = _model(ag, O, material, 419, dev)
        reps = np.full((B, 1), 3)
        a_np = _actions(cloud, B, 1, reps, rng, spread=0.0)
        a_np[:, 0, 0] += 40.0                                    # nobody touches: with sharing only the base rollout runs
        got = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a_np), m, dev, _ppm(task, material))
        ex, need = m.engine(dev).rollout_counts()
        assert need == 3 * B and (ex == 3 if expect else ex == need), (material, B, ex, need)
        assert torch.isfinite(got["state_seqs"]).all()


def test_base_rollout_is_kept_across_calls_with_the_same_start_state(ag, O, dev):
    """The reference's planner calls dynamics() 40 times per planner call with one start state (plan.py:241-247).  In automatic
    mode the base rollout of the prefix sharing stays in the context and is re-used while the start state (compared bit for
    bit on the device), the weights and the task scalars are the same - and recomputed as soon as one of them is not."""
    rng = np.random.default_rng(431)
    task = _task("rope", max_nR=40000)
    W, m = _model(ag, O, "rope", 431, dev)
    cloud = _rope(600, rng)
    B = 96
    eng = m.engine(dev)
    ppm = _ppm(task, "rope")
    reps = rng.integers(2, 7, (B, 1))
    R = int(reps.max())
    a = torch.from_numpy(_actions(cloud, B, 1, reps, rng, spread=3.0))
    s0 = torch.from_numpy(cloud).to(dev)

    def run(state, actions=a):
        out = ag.dynamics(state, actions, m, dev, ppm)["state_seqs"]
        return out, eng.rollout_counts()[0]

    first, ex1 = run(s0)
    with eng.options(share_prefix=0):
        plain, ex0 = run(s0)
    assert torch.equal(first, plain) and ex1 < ex0
    again, ex2 = run(s0.clone())                                  # another buffer, the same bits: the kept base rollout serves it
    assert torch.equal(again, first) and ex2 == ex1 - R, (ex1, ex2, R)
    a2 = torch.from_numpy(_actions(cloud, B, 1, reps, rng, spread=3.0))
    other, ex3 = run(s0, a2)                                      # other pushes, same start state: still served
    with eng.options(share_prefix=0):
        assert torch.equal(other, run(s0, a2)[0])
    moved = s0.clone()
    moved[7, 0] += 1e-6                                           # one ulp-ish change of one coordinate: recomputed
    got, ex4 = run(moved)
    with eng.options(share_prefix=0):
        assert torch.equal(got, run(moved)[0])
    assert ex4 >= ex2 + R - 2 and not torch.equal(got, first)
    back, ex5 = run(s0)                                           # and the old one is gone (one slot): recomputed again
    assert torch.equal(back, first) and ex5 == ex1
    W2 = O.random_weights(432)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W2.items()})
    neww, _ = run(s0)                                             # new weights: the kept rollout must not be used
    with eng.options(share_prefix=0):
        assert torch.equal(neww, run(s0)[0])
    assert not torch.equal(neww, first)


def test_prefix_sharing_in_bf16x3_mode_with_per_particle_physics_and_on_overflow(ag, O, dev):
    rng = np.random.default_rng(443)
    task = _task("rope", max_nR=40000)
    W, m = _model(ag, O, "rope", 443, dev)
    cloud = _rope(200, rng)
    B = 80
    reps = rng.integers(1, 6, (B, 1))
    a = torch.from_numpy(_actions(cloud, B, 1, reps, rng, spread=2.5))
    s0 = torch.from_numpy(cloud).to(dev)
    eng = m.engine(dev)
    ppm = _ppm(task, "rope")

    def both(ppm_, **kw):
        with eng.options(share_prefix=1):
            x = ag.dynamics(s0, a, m, dev, ppm_, **kw)["state_seqs"]
            ex, need = eng.rollout_counts()
        with eng.options(share_prefix=0):
            y = ag.dynamics(s0, a, m, dev, ppm_, **kw)["state_seqs"]
        assert torch.equal(x, y) and ex < need
        return x

    fp32 = both(ppm)
    m.set_precision("bf16x3")
    try:
        b3 = both(ppm)
    finally:
        m.set_precision("fp32")
    assert float((b3 - fp32).abs().max()) <= 1e-4 and not torch.equal(b3, fp32)
    pp = both(ppm, physics_param={"rope": torch.from_numpy(rng.uniform(0.2, 0.8, 200).astype(np.float32))})   # (N_o,) per particle
    assert not torch.equal(pp, fp32)
    tight = _ppm(dict(task, max_nR=500), "rope")                  # the base graph alone exceeds it: the reference raises (utils.py:63-65)
    for sp in (1, 0):
        with eng.options(share_prefix=sp):
            with pytest.raises(Exception, match="Exceeds max dims"):
                ag.dynamics(s0, a, m, dev, tight)
    assert torch.equal(both(ppm), fp32)                           # the context is still good


def test_a_device_planned_call_can_be_captured_in_a_hip_graph(ag, O, dev):
    """A caller may capture an asynchronous dynamics() call (GPU-resident actions, _sync=False) into a hipGraph: the call must then
    neither wait nor look at the host side of an event - no polling of the plan's maxima, no prefix sharing - and a replay must
    reproduce the eager result bit for bit (tools/graph_replay.py measures the latency of exactly this)."""
    rng = np.random.default_rng(457)
    task = _task("rope", max_nR=40000, action_lower_lim=[-4.5, -2.5, -3.14, 0.0], action_upper_lim=[0.0, 4.5, 3.14, 6.0])
    W, m = _model(ag, O, "rope", 457, dev)
    cloud = _rope(150, rng)
    s0 = torch.from_numpy(cloud).to(dev)
    ppm = _ppm(task, "rope")
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    for B in (1, 96):                                            # latency-mode single graph; a batch the prefix sharing would take eagerly
        a = torch.from_numpy(_actions(cloud, B, 1, rng.integers(1, 6, (B, 1)), rng, spread=2.0)).to(dev)
        call = lambda: ag.dynamics(s0, a, m, dev, ppm, _sync=False, _overflow_flag=flag)["state_seqs"]
        with m.engine(dev).options(share_prefix=1 if B > 1 else -1):
            ref = call().clone()
            torch.cuda.synchronize()                             # (one stream per context at a time: include/adaptigraph_amd.h)
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                for _ in range(2):
                    call()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                got = call()
            got.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(got, ref), B
            assert torch.equal(call(), ref)                      # and the context is as good as before
            torch.cuda.synchronize()


def test_rollout_calls_alternating_between_two_streams_equal_the_synchronous_results(ag, O, dev):
    """The workspace and the plans of a call belong to the call slot of its caller stream (r05; r04 serialised calls of different
    streams behind an end-of-call event): back-to-back asynchronous calls alternating between two streams, no synchronisation in
    between, run side by side - every result equals the synchronous one (include/adaptigraph_amd.h; more streams than slots:
    tests/test_gpu_call_slots.py)."""
    rng = np.random.default_rng(461)
    task = _task("rope", max_nR=40000, action_lower_lim=[-4.5, -2.5, -3.14, 0.0], action_upper_lim=[0.0, 4.5, 3.14, 6.0])
    W, m = _model(ag, O, "rope", 461, dev)
    cloud = _rope(150, rng)
    s0 = torch.from_numpy(cloud).to(dev)
    ppm = _ppm(task, "rope")
    flag = torch.zeros(4, dtype=torch.int32, device=dev)
    acts = [torch.from_numpy(_actions(cloud, 96, 1, rng.integers(1, 6, (96, 1)), rng, spread=2.0)).to(dev) for _ in range(6)]
    want = [ag.dynamics(s0, a, m, dev, ppm)["state_seqs"].clone() for a in acts]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    got = []
    for i, a in enumerate(acts):
        with torch.cuda.stream(streams[i % 2]):
            got.append(ag.dynamics(s0, a, m, dev, ppm, _sync=False, _overflow_flag=flag)["state_seqs"])
    torch.cuda.synchronize()
    for i in range(len(acts)):
        assert torch.equal(got[i], want[i]), i
