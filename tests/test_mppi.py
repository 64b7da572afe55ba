"""MPPI sampling / update kernels (csrc/ag_mppi.hip) against vectors recorded from the reference
(tests/golden/mppi.npz: plan_utils.py:31-101 run with seeded torch, its random draws recorded alongside).

Tolerance: the device evaluates cos / sin / atan2 / exp / sqrt with its own fp32 routines and sums the softmax in a
fixed tree instead of torch's order, so values agree to a few ulp: |err| <= 1e-6 + 1e-6*|value| (lengths reach 10).
Uniform resampling (one multiply-add per component) and clipping of in-range values are bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden

ATOL, RTOL = 1e-6, 1e-6


def _close(got, want):
    return np.all(np.abs(got - want) <= ATOL + RTOL * np.abs(want)), float(np.abs(got - want).max())


def test_golden_draws_reproduce_the_reference_samples_on_cpu():
    """The fixture's recorded draws are the ones behind its samples (CPU, torch only): pins the draw contract the device
    sampler relies on - torch.rand((S,H,4)) once, or torch.normal(0, noise_level, (S,4)) once per look-ahead step."""
    g = load_golden("mppi")
    lo, hi = torch.from_numpy(g["lo"]), torch.from_numpy(g["hi"])
    torch.manual_seed(7)
    assert np.array_equal(torch.rand((64, 3, 4)).numpy(), g["uniform_iter0"])
    assert np.array_equal((torch.from_numpy(g["uniform_iter0"]) * (hi - lo) + lo).numpy(), g["sample_iter0"])
    torch.manual_seed(8)
    assert np.array_equal(torch.stack([torch.normal(0, 0.3, (64, 4)) for _ in range(3)]).numpy(), g["noise_iter1"])
    assert np.array_equal(g["sample_iter1"][0], g["act_seq"])          # sample 0 keeps the nominal sequence (:75)


@pytest.mark.gpu
def test_mppi_kernels_vs_reference_golden():
    import adaptigraph_amd as ag
    dev = torch.device("cuda:0")
    g = load_golden("mppi")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    lo, hi, act_seq = t("lo"), t("hi"), t("act_seq")
    s0 = ag.sample_action_seq(act_seq, lo, hi, 64, dev, iter_index=0, noise_level=1.0, push_length=0.1,
                              _draws=g["uniform_iter0"])
    assert np.array_equal(s0.cpu().numpy(), g["sample_iter0"])         # u*(hi-lo)+lo: bit-exact
    s1 = ag.sample_action_seq(act_seq, lo, hi, 64, dev, iter_index=1, noise_level=0.3, push_length=0.1,
                              _draws=g["noise_iter1"])
    ok, err = _close(s1.cpu().numpy(), g["sample_iter1"])
    assert ok, err
    assert torch.equal(s1[0], act_seq)                                  # nominal action untouched
    up = ag.optimize_action_mppi(t("sample_iter1"), t("rewards"), reward_weight=500.0, action_lower_lim=lo,
                                 action_upper_lim=hi, push_length=0.1)
    ok, err = _close(up.cpu().numpy(), g["mppi"])
    assert ok, err
    cl = ag.clip_actions(t("wild"), lo, hi).cpu().numpy()
    ok, err = _close(cl, g["clipped"])
    assert ok, err
    th = torch.linspace(-20, 20, 101, device=dev)
    want = ((th.cpu().double() + np.pi) % (2 * np.pi)) - np.pi
    assert float((ag.angle_normalize(th).cpu().double() - want).abs().max()) < 5e-6


@pytest.mark.gpu
def test_mppi_sampler_seeded_like_the_reference_and_update_properties():
    """Without `_draws` the sampler draws with the reference's torch calls on the device; the update is invariant to a
    permutation of the candidates up to summation order and reduces to the best candidate for a huge reward weight."""
    import adaptigraph_amd as ag
    dev = torch.device("cuda:0")
    g = load_golden("mppi")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    lo, hi, act_seq = t("lo"), t("hi"), t("act_seq")
    torch.manual_seed(3)
    a = ag.sample_action_seq(act_seq, lo, hi, 4096, dev, iter_index=1, noise_level=0.3, push_length=0.1)
    torch.manual_seed(3)
    noise = torch.stack([torch.normal(0, 0.3, (4096, 4), device=dev) for _ in range(3)])
    b = ag.sample_action_seq(act_seq, lo, hi, 4096, dev, iter_index=1, noise_level=0.3, push_length=0.1, _draws=noise)
    assert torch.equal(a, b)
    assert bool(((a >= lo - 1e-6) & (a <= hi + 1e-6))[1:].all())
    torch.manual_seed(4)
    u = ag.sample_action_seq(act_seq, lo, hi, 4096, dev, iter_index=0)
    assert bool(((u >= lo) & (u <= hi)).all()) and float(u[..., 0].std()) > 0.5
    rew = torch.randn(4096, device=dev) * 0.02 - 5.0
    up = ag.optimize_action_mppi(a, rew, 500.0, lo, hi, 0.1)
    perm = torch.randperm(4096, device=dev)
    up_p = ag.optimize_action_mppi(a[perm], rew[perm], 500.0, lo, hi, 0.1)
    assert float((up - up_p).abs().max()) < 1e-4
    best = int(torch.argmax(rew))
    sharp = ag.optimize_action_mppi(a, rew, 1e9, lo, hi, 0.1)           # softmax collapses onto the best candidate
    want = ag.clip_actions(a[best], lo, hi)
    assert float((sharp - want).abs().max()) < 1e-4


@pytest.mark.gpu
def test_nan_rewards_and_actions_surface_as_nan_like_the_reference():
    """The reference's softmax / clamp_ propagate NaN (plan_utils.py:31-39, 83): one NaN reward makes the whole updated
    action NaN; a NaN component of an action stays NaN through clip_actions.  A device clamp spelled fmin(fmax()) would
    return the lower limit instead - a valid-looking action."""
    import adaptigraph_amd as ag
    dev = torch.device("cuda:0")
    g = load_golden("mppi")
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    lo, hi = t("lo"), t("hi")
    nan = float("nan")
    assert torch.isnan(torch.clamp(torch.tensor([nan]), -1.0, 1.0)).all()              # the reference's semantics (CPU torch)
    assert torch.isnan(torch.softmax(torch.tensor([0.0, nan, 1.0]), 0)).all()
    rew = t("rewards").clone()
    rew[17] = nan
    up = ag.optimize_action_mppi(t("sample_iter1"), rew, 500.0, lo, hi, 0.1)
    assert torch.isnan(up).all()
    wild = t("wild").clone()
    wild[1, 2, 0] = nan
    wild[2, 0, 2] = nan
    cl = ag.clip_actions(wild, lo, hi)
    want = torch.from_numpy(g["clipped"]).to(dev)
    bad = torch.zeros_like(wild, dtype=torch.bool)
    bad[1, 2, 0] = True
    bad[2, 0, 2] = True
    assert torch.isnan(cl[bad]).all() and not torch.isnan(cl[~bad]).any()
    assert float((cl[~bad] - want[~bad]).abs().max()) <= 1e-5
