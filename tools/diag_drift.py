"""Diagnostic (GPU box): where does a free-running rollout leave the oracle?  Re-creates the cloth 1024 x 20 case of
tests/test_gpu_fullsize.py, rolls the chosen candidates with repeat = 1..R (same push, so candidate k holds the state
after k steps), and compares every step with the oracle's trace: position error, edge-set differences (free-running and
teacher-forced on the oracle's own positions)."""
import sys, os, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adaptigraph_amd as ag
from oracle import adaptigraph_oracle as O
from test_gpu_more import _task, _grid, _actions, _model
from test_gpu_parity import _ppm

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 43
rng = np.random.default_rng(seed)
task = _task("cloth", max_nR=40000)
W, m = _model(ag, O, "cloth", seed, dev)
cloud = _grid(45, 0.3, 0.02, rng)
B = 1024
a_np = _actions(cloud, B, 2, 10, rng, spread=1.5)
N_o = cloud.shape[0]
for cand in (0, B - 1):
    act = a_np[cand]
    tr = []
    want = O.dynamics(W, 3, cloud, act[None], task, trace=tr)["state_seqs"][0]
    tr = tr[0]                                             # 20 records
    # GPU: first look-ahead step with repeat k = 1..10
    a1 = np.repeat(act[None, :1], 10, 0).copy()
    a1[:, 0, 3] = np.arange(1, 11) + 0.5
    g1 = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a1).to(dev), m, dev, _ppm(task, "cloth"))["state_seqs"][:, 0].cpu().numpy()
    a2 = np.repeat(act[None], 10, 0).copy()
    a2[:, 1, 3] = np.arange(1, 11) + 0.5
    g2 = ag.dynamics(torch.from_numpy(cloud).to(dev), torch.from_numpy(a2).to(dev), m, dev, _ppm(task, "cloth"))["state_seqs"][:, 1].cpu().numpy()
    gpu_steps = np.concatenate([g1, g2], 0)                # (20, N_o, 3): state after step k+1
    for k in range(20):
        rec = tr[k]
        err = np.abs(gpu_steps[k] - rec["pred_pos"]).max()
        # edges the oracle used at this step vs GPU builder on the oracle's positions (teacher-forced) and on the GPU's own
        pos_o = rec["state_last"]
        N = pos_o.shape[0]
        mask = torch.ones((1, N), dtype=torch.bool, device=dev); tool = torch.zeros((1, N), dtype=torch.bool, device=dev); tool[:, N_o:] = True
        el = ag.construct_edges_index(torch.from_numpy(pos_o[None]).to(dev), task["adj_thresh"], mask, tool, task["topk"], True)
        n = int(el.n_edges[0]); r = el.recv[0, :n].cpu().numpy(); s = el.send[0, :n].cpu().numpy()
        tf_same = len(r) == len(rec["recv"]) and np.array_equal(r, rec["recv"]) and np.array_equal(s, rec["send"])
        # GPU's own positions at this step: previous GPU state + tool from oracle (tool pos is same up to y)
        if k > 0 and k != 10:
            pos_g = pos_o.copy(); pos_g[:N_o] = gpu_steps[k - 1]
            el2 = ag.construct_edges_index(torch.from_numpy(pos_g[None]).to(dev), task["adj_thresh"], mask, tool, task["topk"], True)
            n2 = int(el2.n_edges[0])
            eo = set(zip(rec["recv"].tolist(), rec["send"].tolist()))
            eg = set(zip(el2.recv[0, :n2].cpu().numpy().tolist(), el2.send[0, :n2].cpu().numpy().tolist()))
            diff = (eo - eg, eg - eo)
        else:
            diff = (set(), set())
        print(f"cand {cand} step {k+1:2d}: |gpu-oracle| {err:.2e}  teacher-forced edges equal {tf_same}  free-running edge diff -{len(diff[0])} +{len(diff[1])} {sorted(diff[0])[:3]} {sorted(diff[1])[:3]}")
        if diff[0] or diff[1]:
            for (i, j) in sorted(diff[0] | diff[1])[:3]:
                d_o = np.float32(((pos_o[i] - pos_o[j]) ** 2).sum()); d_g = np.float32(((pos_g[i] - pos_g[j]) ** 2).sum())
                print(f"      pair ({i},{j}): dis oracle {d_o:.9f} gpu {d_g:.9f}  thr2 {np.float32(task['adj_thresh'])**2:.9f}")
