"""decode_action with the reference's exact arithmetic (src/planning/plan_utils.py:11-20)."""
import torch


def decode_action(action, push_length=0.10):
    x_start = action[:, :, 0]
    z_start = action[:, :, 1]
    theta = action[:, :, 2]
    length = action[:, :, 3].detach()
    action_repeat = length.to(torch.int32)
    x_end = x_start - push_length * torch.cos(theta)
    z_end = z_start - push_length * torch.sin(theta)
    decoded_action = torch.stack([x_start, z_start, x_end, z_end], dim=-1)
    return decoded_action, action_repeat
