"""The C boundary used from plain C (-m gpu): tools/abi_client/abi_client.c - C99, no Python, no C++, no torch in the process -
is compiled with gcc against include/adaptigraph_amd.h and linked with libadaptigraph_hip.so + the HIP runtime, runs one
rollout through ag_rollout (host-decoded actions) and one through ag_rollout_actions (raw actions on the device), and its
outputs must equal what the Python shim gets for the same case: bit for bit on the host-decoded path, and on the
device-planned path too (same kernel, same inputs)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from test_gpu_parity import _ppm
from test_gpu_more import _task, _grid, _actions, _model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["particle_encoder.model.0", "particle_encoder.model.2", "particle_encoder.model.4",
         "relation_encoder.model.0", "relation_encoder.model.2", "relation_encoder.model.4",
         "particle_propagator.linear", "relation_propagator.linear",
         "non_rigid_predictor.linear_0", "non_rigid_predictor.linear_1", "non_rigid_predictor.linear_2"]


def _build(tmp):
    exe = os.path.join(tmp, "abi_client")
    csrc = os.path.join(ROOT, "adaptigraph_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tools", "abi_client", "abi_client.c"), "-L", csrc, "-ladaptigraph_hip",
           "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_client_matches_the_python_shim(tmp_path):
    import adaptigraph_amd as ag
    from adaptigraph_amd.forward_dynamics import _tool_layout
    from adaptigraph_amd.plan_utils import decode_action
    from oracle import adaptigraph_oracle as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(113)
    task = _task("granular", max_nR=20000, action_upper_lim=[0.0, 4.5, 3.14, 6.0])       # 5-point pusher
    W, m = _model(ag, O, "granular", 113, dev)
    cloud = _grid(12, 0.12, 0.02, rng)
    B, H = 9, 2
    reps = rng.integers(1, 6, (B, H))
    a_np = _actions(cloud, B, H, reps, rng, spread=0.4)
    act = torch.from_numpy(a_np)
    decoded, repeat = decode_action(act, push_length=task["push_length"])
    xz, delta = _tool_layout(decoded, act[:, :, 2], task)
    M = task["eef_num"]
    hdr = struct.pack("12i", 0x41474331, B, H, cloud.shape[0], M, task["topk"], int(task["connect_tools_all"]), task["max_nR"],
                      int(task["gripper_enable"]), 3, 4, 6)
    offs = [0.0] + [float(task["pusher_points"][k][1]) * task["sim_real_ratio"] for k in range(1, M)] + [0.0] * (8 - M)
    fl = struct.pack("12f", task["adj_thresh"], 0.0, 0.5, task["push_length"], *offs)
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(hdr + fl)
        for base in ORDER:
            for suffix in (".weight", ".bias"):
                f.write(np.ascontiguousarray(W[base + suffix], np.float32).tobytes())
        for arr in (cloud, xz.numpy(), delta.numpy(), a_np):
            f.write(np.ascontiguousarray(arr, np.float32).tobytes())
        f.write(np.ascontiguousarray(repeat.numpy(), np.int32).tobytes())
    exe = _build(str(tmp_path))
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(case), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_client ok" in r.stdout, r.stdout + r.stderr
    raw = open(out, "rb").read()
    n = B * H * cloud.shape[0] * 3
    got1 = np.frombuffer(raw, np.float32, n, 0).reshape(B, H, -1, 3)
    got2 = np.frombuffer(raw, np.float32, n, 4 * n).reshape(B, H, -1, 3)
    dec2 = np.frombuffer(raw, np.float32, B * H * 4, 8 * n).reshape(B, H, 4)
    ex, need = np.frombuffer(raw, np.int64, 2, 8 * n + 16 * B * H)
    assert ex == need == int(reps.sum())
    s0 = torch.from_numpy(cloud).to(dev)
    eng = m.engine(dev)
    with eng.options(device_decode=0):
        host = ag.dynamics(s0, act.to(dev), m, dev, _ppm(task, "granular"))
    assert np.array_equal(got1, host["state_seqs"].cpu().numpy())                        # the same C calls, the same bits
    devp = ag.dynamics(s0, act.to(dev), m, dev, _ppm(task, "granular"))
    assert np.array_equal(got2, devp["state_seqs"].cpu().numpy())
    assert np.array_equal(dec2, devp["action_seqs"].cpu().numpy())
    want = O.dynamics(W, 3, cloud, a_np, task)["state_seqs"]
    assert np.abs(got1 - want).max() <= 1e-5
