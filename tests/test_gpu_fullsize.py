"""GPU parity at the sizes that are benchmarked (-m gpu): every BASELINE.json config at its own batch x horizon, and
the code path the headline number is measured on (two in-library streams, several launch chunks per stream,
second workspace carve, self-loop constant rows behind the last chunk) against the oracle.

The dense reference cannot run these sizes; the numpy oracle (pinned on the reference's goldens,
tests/test_oracle_vs_golden.py) rolls out a few candidates of each batch, and size-independent properties cover the
rest: candidates are independent, so any sub-batch, chunking or stream count must give the same bits per candidate.
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


class _env:
    """os.environ override that the C side sees through getenv (AG_STREAMS is read on every call)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


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
    with _env(AG_STREAMS=2):
        two = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    assert torch.isfinite(two).all()
    with _env(AG_STREAMS=1):
        one = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        eng.set_chunk(37)                                               # odd chunk size, short last chunk, one stream
        odd = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
        eng.set_chunk(0)
    assert torch.equal(two, one)
    assert torch.equal(two, odd)
    picks = sorted({b for lo, hi in chunks for b in (lo, hi - 1)})
    small = ag.dynamics(s0, a[picks], m, dev, ppm)["state_seqs"]        # 8 candidates: one stream, one chunk
    assert torch.equal(two[picks], small)
    want = O.dynamics(W, 3, cloud, a_np[picks], task)["state_seqs"]
    err = np.abs(two[picks].cpu().numpy() - want).max()
    print(f"two-stream path, chunks {chunks}: candidates {picks} vs oracle {err:.2e}")
    assert err <= POS_TOL, err


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
    seq = ag.dynamics(s0, a, m, dev, ppm)["state_seqs"]
    assert seq.shape == (B, 2, cloud.shape[0], 3) and torch.isfinite(seq).all()
    _property_checks(ag, m, dev, s0, a, ppm, seq)
    picks = [0, B - 1]
    want = O.dynamics(W, 3, cloud, a_np[picks], task)["state_seqs"]     # 2 candidates x 20 free-running steps
    err = np.abs(seq[picks].cpu().numpy() - want).max()
    print(f"{material} {B} x 20: candidates {picks} vs oracle {err:.2e}")
    assert err <= POS_TOL, err


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
    want = O.dynamics_masked(W, 3, state[picks], mask[picks], a_np[picks], task)["state_seqs"]
    got = seq[picks].cpu().numpy()
    valid = mask[picks]
    err = np.abs(got - want)[valid].max()
    print(f"cfg4 {material}: counts {counts[picks].tolist()} vs oracle {err:.2e}")
    assert err <= POS_TOL, err
