"""Action codec of the planner: replaces decode_action (reference src/planning/plan_utils.py:11-20).

An action is (x, z, theta, length): a push that starts at (x, z), heads AGAINST the direction theta for `push_length`
per repeat, and is applied int(length) times: end = start - push_length * (cos theta, sin theta), each product and
difference in the reference's order.  This function runs torch ops on whatever device `action` is on.  dynamics() calls it
on the HOST copy of the actions (torch CPU cos/sin: bit-equal to a CPU reference) unless the actions are GPU-resident and
the device-planned path applies - then the same arithmetic runs inside csrc/ag_graph.hip: k_roll_plan with the device's
cos/sin (forward_dynamics.py docstring), and this function is not involved.
"""
import torch


def decode_action(action, push_length=0.10):
    """action (..., 4) -> decoded (..., 4) = [x_start, z_start, x_end, z_end], action_repeat (...,) int32 (truncation)."""
    start = action[..., 0:2]
    heading = action[..., 2]
    reach = push_length * torch.stack((torch.cos(heading), torch.sin(heading)), dim=-1)
    decoded = torch.cat((start, start - reach), dim=-1)
    repeat = action[..., 3].detach().to(torch.int32)
    return decoded, repeat
