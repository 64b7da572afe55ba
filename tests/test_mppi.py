"""SURVEY §8(f) rank 2 on CPU: MPPI sampling / update against vectors from the reference (same torch, same seeds)."""
import numpy as np
import torch

from helpers import load_golden
from adaptigraph_amd.mppi import sample_action_seq, optimize_action_mppi, clip_actions


def test_mppi_functions_match_reference_bitwise():
    g = load_golden("mppi")
    lo, hi = torch.from_numpy(g["lo"]), torch.from_numpy(g["hi"])
    act_seq = torch.from_numpy(g["act_seq"])
    torch.manual_seed(7)
    s0 = sample_action_seq(act_seq, lo, hi, 64, torch.device("cpu"), iter_index=0, noise_level=1.0, push_length=0.1)
    torch.manual_seed(8)
    s1 = sample_action_seq(act_seq, lo, hi, 64, torch.device("cpu"), iter_index=1, noise_level=0.3, push_length=0.1)
    assert np.array_equal(s0.numpy(), g["sample_iter0"]) and np.array_equal(s1.numpy(), g["sample_iter1"])
    assert torch.equal(s1[0], act_seq)                    # sample 0 keeps the nominal sequence (plan_utils.py:75)
    up = optimize_action_mppi(s1, torch.from_numpy(g["rewards"]), reward_weight=500.0, action_lower_lim=lo,
                              action_upper_lim=hi, push_length=0.1)
    assert np.array_equal(up.numpy(), g["mppi"])
    assert np.array_equal(clip_actions(torch.from_numpy(g["wild"]), lo, hi).numpy(), g["clipped"])
