"""Throughput of the BASELINE configs that are parity cases, not the bench line (BASELINE.json configs[0..2, 4]):
rope 1 x 10, rope 64 x 20, granular 256 x 20, mixed 512 x 20 (variable-size graphs through dynamics_masked-style
padding, one model context per material).  Synthetic clouds as SURVEY 8(d); random-init weights.  One GPU.
Prints one JSON object per config.  Diagnostic: the contract line is bench.py's."""
import json, os, sys, time, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                            # random_weights, model_cfg shapes
import adaptigraph_amd as ag

dev = torch.device("cuda", 0)
TASKS = {
    "rope": dict(adj_thresh=0.5, topk=10, connect_tools_all=False, push_length=0.1, gripper_enable=False, eef_num=1,
                 pusher_points=[[0.0, 0.0, 0.12]]),
    "granular": dict(adj_thresh=0.4, topk=20, connect_tools_all=False, push_length=0.2, gripper_enable=False, eef_num=5,
                     pusher_points=[[0, 0, 0.1], [0, 0.05, 0.1], [0, 0.025, 0.1], [0, -0.025, 0.1], [0, -0.05, 0.1]]),
    "cloth": dict(adj_thresh=0.75, topk=5, connect_tools_all=True, push_length=0.1, gripper_enable=True, eef_num=1,
                  pusher_points=[[0.0, 0.0, 0.17]]),
}


def task_of(mat, N_o):
    t = dict(sim_real_ratio=10, max_n=1, n_his=4, material=mat, material_dims={mat: 1}, material_indices={mat: 0})
    t.update(TASKS[mat])
    t["max_nR"] = int(1.2 * (t["topk"] + t["eef_num"]) * (N_o + t["eef_num"])) + 64
    return t


def cloud_of(mat, rng):
    if mat == "rope":
        t = np.linspace(0, 1, 300)
        p = np.stack([-2 + 3 * t, 0 * t, 0.5 * np.sin(6 * t)], 1)
        return (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)
    side, pitch, jit = (32, 0.12, 0.02) if mat == "granular" else (45, 0.3, 0.02)
    g = (np.arange(side) - (side - 1) / 2.0) * pitch
    xx, zz = np.meshgrid(g, g, indexing="ij")
    p = np.stack([xx.ravel() - 2.0, np.zeros(side * side), zz.ravel() + 1.0], 1)
    return (p + rng.normal(0, jit, p.shape)).astype(np.float32)


def model_of(mat):
    mc, _, _ = B.model_cfg()
    m = ag.DynamicsPredictor(mc, {"material_index": {mat: 0}, mat: {"physics_params": [{"name": "p", "use": True}]}},
                             {"n_his": 4, "materials": [mat]}, dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in B.random_weights(0).items()})
    return m


def ppm_of(task, mat):
    return types.SimpleNamespace(task_config=task, eef_num=task["eef_num"], material=mat, material_dims=task["material_dims"],
                                 material_indices=task["material_indices"], physics_param={mat: torch.tensor([0.5])},
                                 adj_thresh=task["adj_thresh"])


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


FAMILIES = ["edge_count", "edge_emit", "node_enc", "edge_enc", "node_prop", "node_final", "roll_init", "roll_update"]


def start_graph_edges(task, clouds, masks, eef_xz):
    """(edges, encoded = non-self-loop edges) of the START graphs of a batch: clouds (B,N_o,3), masks (B,N_o) bool,
    eef_xz (B,M,2) tool keypoints; the tool sits at the cloud's lowest valid y (forward_dynamics.py:40).  The counts move by a
    few edges over a rollout; they price the edge chain's algorithmic FLOPs (140,100 per encoded edge)."""
    Bn, N_o, M = clouds.shape[0], clouds.shape[1], eef_xz.shape[1]
    y = np.where(masks, clouds[..., 1], np.inf).min(1)
    tool = np.stack([eef_xz[..., 0], np.repeat(y[:, None], M, 1), eef_xz[..., 1]], -1).astype(np.float32)
    pos = torch.from_numpy(np.concatenate([clouds, tool], 1)).to(dev)
    mask = torch.from_numpy(np.concatenate([masks, np.ones((Bn, M), bool)], 1)).to(dev)
    tmask = torch.zeros((Bn, N_o + M), dtype=torch.bool, device=dev)
    tmask[:, N_o:] = True
    E = Enc = 0
    for b0 in range(0, Bn, 64):
        el = ag.construct_edges_index(pos[b0:b0 + 64], task["adj_thresh"], mask[b0:b0 + 64], tmask[b0:b0 + 64], task["topk"],
                                      task["connect_tools_all"])
        n = el.n_edges.long()
        live = torch.arange(el.edge_cap, device=dev)[None, :] < n[:, None]
        E += int(n.sum())
        Enc += int(n.sum()) - int(((el.recv == el.send) & live).sum())
    return E, Enc


def kernel_report(engines, fn, n_enc_edge_forwards, n_node_forwards, wall_ms, n_edge_forwards=None):
    """One more call with HIP events around every launch (engine pinned to one stream): ms per family, the dominant kernel's
    algorithmic rate against the fp32 MFMA peak, and the edge builder's share of the single-stream kernel time."""
    # (both sharings off for this pass: every forward then encodes every edge, which is what the algorithmic FLOP counts assume)
    old = [(e.get_option("share_first"), e.get_option("share_prefix")) for e in engines]
    for e in engines:
        e.set_option("share_first", 0); e.set_option("share_prefix", 0)
        e.reset_stats(); e.set_profiling(FAMILIES)
    fn(); torch.cuda.synchronize()
    for e, (sf, sp) in zip(engines, old):
        e.set_option("share_first", sf); e.set_option("share_prefix", sp)
    fam = {f: [0.0, 0] for f in FAMILIES}
    for e in engines:
        for f in FAMILIES:
            ms, n = e.kernel_stats(f)
            fam[f][0] += ms; fam[f][1] += n
        e.set_profiling([]); e.reset_stats()
    total = sum(v[0] for v in fam.values())
    dom = max(fam, key=lambda f: fam[f][0])
    flop = {"edge_enc": B.FLOP_PER_EDGE * n_enc_edge_forwards, "node_prop": B.FLOP_PER_NODE_PROP * n_node_forwards * 2,
            "node_final": B.FLOP_PER_NODE_FINAL * n_node_forwards}
    rep = {"kernel_ms_single_stream": {f: round(v[0], 3) for f, v in fam.items() if v[1]},
           "launches": {f: v[1] for f, v in fam.items() if v[1]}, "kernel_ms_total": total,
           "dominant_kernel": dom, "dominant_share": fam[dom][0] / total,
           "edge_builder_share": (fam["edge_count"][0] + fam["edge_emit"][0]) / total}
    for f, fl in flop.items():
        if fam[f][0] > 0:
            tf = fl / (fam[f][0] * 1e-3) / 1e12
            rep[f] = {"algorithmic_flop": fl, "ms": fam[f][0], "tflops": tf, "frac_of_fp32_mfma_peak": tf / B.PEAK_FP32_MFMA_TFLOPS}
    # the propagate chains are bounded by both resources: compulsory HBM bytes per round = one 640-B C row per encoded edge,
    # a 4-B index per edge, and six 640-B rows per particle (U, V, eff in; eff, U, V out) - the final round three
    if n_edge_forwards is not None:
        per_round = n_enc_edge_forwards * 640 + n_edge_forwards * 4
        for f, rounds, rows in (("node_prop", 2, 6), ("node_final", 1, 3)):
            if f in rep:
                by = rounds * (per_round + n_node_forwards * 640 * rows)
                gbs = by / (fam[f][0] * 1e-3) / 1e9
                rep[f].update(compulsory_hbm_bytes=by, hbm_gbs=gbs, frac_of_hbm_peak=gbs / B.PEAK_HBM_GBS)
    f_exec = sum(flop.values())
    rep["end_to_end_frac_of_fp32_mfma_peak"] = f_exec / (wall_ms * 1e-3) / 1e12 / B.PEAK_FP32_MFMA_TFLOPS
    return rep


def homogeneous(mat, Bn, H, R, reps, report=True):
    rng = np.random.default_rng(0)
    cloud = cloud_of(mat, rng)
    task = task_of(mat, cloud.shape[0])
    task["action_upper_lim"] = [0.0, 4.5, 3.14, float(R)]          # bounds the repeat: GPU-resident actions take ag_rollout_actions
    m, s0 = model_of(mat), torch.from_numpy(cloud).to(dev)
    a_np = B.make_actions(Bn, H, R, cloud, rng)
    a = torch.from_numpy(a_np).to(dev)
    ppm = ppm_of(task, mat)
    fn = lambda: ag.dynamics(s0, a, m, dev, ppm)
    dt = timed(fn, reps)
    out = {"config": f"{mat} {cloud.shape[0]}+{task['eef_num']} particles, {Bn} candidates x {H * R} steps",
           "ms_per_call": dt * 1e3, "rollout_steps_per_s": Bn * H * R / dt}
    if report:
        N_o, M = cloud.shape[0], task["eef_num"]
        dec, _ = ag.decode_action(torch.from_numpy(a_np[:, :1]), push_length=task["push_length"])
        from adaptigraph_amd.forward_dynamics import _tool_layout
        xz, _ = _tool_layout(dec, torch.from_numpy(a_np[:, :1, 2]), task)
        E, Enc = start_graph_edges(task, np.repeat(cloud[None], Bn, 0), np.ones((Bn, N_o), bool), xz[:, 0].numpy())
        out.update(edges_per_graph=E / Bn, edges_encoded_per_graph=Enc / Bn)
        out.update(kernel_report([m.engine(dev)], fn, Enc * H * R, Bn * (N_o + M) * H * R, dt * 1e3, E * H * R))
    return out


def mixed(total, steps, reps, report=True):
    """cfg 5: a third of the batch per material, every candidate with its own particle count U{N/2..N} (padded + masked);
    dynamics_masked advances one look-ahead step of `steps` repeats."""
    rng = np.random.default_rng(1)
    calls, batches, n_steps, engines, enc_fwd, node_fwd, edge_fwd = [], [], 0, [], 0, 0, 0
    for mat, nb in (("rope", total // 3 + total % 3), ("granular", total // 3), ("cloth", total // 3)):
        cloud = cloud_of(mat, rng)
        N = cloud.shape[0]
        task = task_of(mat, N)
        m, ppm = model_of(mat), ppm_of(task_of(mat, N), mat)
        state = np.repeat(cloud[None], nb, 0)
        mask = np.zeros((nb, N), bool)
        for b in range(nb):
            mask[b, :rng.integers(N // 2, N + 1)] = True
        a = B.make_actions(nb, 1, steps, cloud, rng)[:, 0]
        args = (torch.from_numpy(state).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(a).to(dev))
        calls.append(lambda args=args, m=m, ppm=ppm: ag.dynamics_masked(*args, m, dev, ppm))
        batches.append((args[0], args[1], torch.from_numpy(a), m, ppm))      # (CPU-resident pushes: no read-back inside the call)
        n_steps += nb * steps
        engines.append(m.engine(dev))
        if report:
            from adaptigraph_amd.forward_dynamics import _tool_layout
            dec, _ = ag.decode_action(torch.from_numpy(a[:, None]), push_length=task["push_length"])
            xz, _ = _tool_layout(dec, torch.from_numpy(a[:, None, 2]), task)
            E, Enc = start_graph_edges(task, state, mask, xz[:, 0].numpy())
            enc_fwd += Enc * steps
            edge_fwd += E * steps
            node_fwd += (int(mask.sum()) + nb * task["eef_num"]) * steps
    fn = lambda: [c() for c in calls]
    fn_mixed = lambda: ag.dynamics_mixed(batches, dev)
    seq_out, mix_out = fn(), fn_mixed()
    same = all(torch.equal(a_["state_seqs"], b_["state_seqs"]) for a_, b_ in zip(seq_out, mix_out))
    dt_seq = timed(fn, reps)
    dt_seq_cpu = timed(lambda: [ag.dynamics_masked(b_[0], b_[1], b_[2], b_[3], dev, b_[4]) for b_ in batches], reps)
    dt_pin = timed(lambda: ag.dynamics_mixed(batches, dev, one_stream_each=True), reps)
    dt_fork = timed(lambda: ag.dynamics_mixed(batches, dev, one_stream_each=False), reps)
    dt_given = timed(lambda: ag.dynamics_mixed(batches, dev, largest_first=False), reps)
    dt_two = timed(lambda: ag.dynamics_mixed(batches, dev, streams_each=2), reps)
    dt = timed(fn_mixed, reps)
    out = {"config": f"mixed rope+granular+cloth, {total} variable-size graphs x {steps} steps", "ms_per_call": dt * 1e3,
           "rollout_steps_per_s": n_steps / dt, "entry": "adaptigraph_amd.dynamics_mixed (three materials dealt to three streams, one read-back)",
           "ms_three_sequential_dynamics_masked_calls": dt_seq * 1e3, "ms_three_sequential_calls_cpu_resident_pushes": dt_seq_cpu * 1e3,
           "ms_dynamics_mixed_engines_on_one_stream_each": dt_pin * 1e3, "ms_dynamics_mixed_engines_fork_by_size": dt_fork * 1e3,
           "ms_dynamics_mixed_in_the_given_order_rope_granular_cloth": dt_given * 1e3, "ms_dynamics_mixed_two_streams_per_engine": dt_two * 1e3,
           "bit_equal_to_the_sequential_calls": bool(same)}
    if report:
        out.update(kernel_report(engines, fn, enc_fwd, node_fwd, dt * 1e3, edge_fwd))
    return out


def ppm_sweep(n_points=50, n_inter=20, reps=3):
    """the physics-parameter optimiser's inner loop (physics_param_optimizer.py:75-122, 178-226) - the objective
    dynamics_error evaluated at 50 physics parameters over 20 past interactions (masked rope clouds of 120..200 particles, pushes
    of the shipped length range).  Sequential calls (what gp_minimize issues: each proposal depends on the last value) against
    dynamics_error_sweep (a CMA-ES population / a sweep: independent evaluations dealt to streams, one read-back)."""
    rng = np.random.default_rng(5)
    mat = "rope"
    t = np.linspace(0, 1, 200)
    p = np.stack([-3.5 + 2.5 * t, 0 * t, 1.0 + 0.5 * np.sin(6 * t)], 1)
    full = (p + rng.normal(0, 0.01, p.shape)).astype(np.float32)
    task = task_of(mat, 200)
    task["max_nobj"] = 200
    m = model_of(mat)
    ppm = ppm_of(task, mat)
    ppm.model, ppm.device = m, dev
    inits, reals, acts = [], [], []
    for i in range(n_inter):
        n = int(rng.integers(120, 201))
        keep = np.sort(rng.choice(200, n, replace=False))
        inits.append(full[keep])
        reals.append((full[keep] + rng.normal(0, 0.02, (n, 3)) + np.float32([0.05, 0, 0.03])).astype(np.float32))
        a = B.make_actions(1, 1, int(rng.integers(5, 15)), full, rng)[0, 0]
        acts.append(a)
    params = [[float(v)] for v in np.linspace(-0.2, 1.2, n_points)]
    seq = lambda: [ag.dynamics_error(v, ppm, inits, reals, acts) for v in params]
    swp = lambda: ag.dynamics_error_sweep(params, ppm, inits, reals, acts)
    a0, a1 = np.asarray(seq(), np.float64), swp()
    assert np.array_equal(a0, a1), float(np.abs(a0 - a1).max())
    out = {"config": f"rope, dynamics_error over {n_points} physics parameters x {n_inter} masked interactions (120..200 of 200 particles, "
                     f"repeats 5..14)", "bit_equal": True}
    for label, fn in (("sequential", seq), ("sweep", swp)):
        dt = timed(fn, reps)
        out[f"ms_per_evaluation_{label}"] = dt * 1e3 / n_points
    for s_ in (1, 2):
        dt = timed(lambda: ag.dynamics_error_sweep(params, ppm, inits, reals, acts, streams=s_), reps)
        out[f"ms_per_evaluation_sweep_{s_}_stream{'s' if s_ > 1 else ''}"] = dt * 1e3 / n_points
    out["speedup_sweep_vs_sequential"] = out["ms_per_evaluation_sequential"] / out["ms_per_evaluation_sweep"]
    # kernel share of one evaluation (HIP events, one stream)
    eng = m.engine(dev)
    eng.reset_stats(); eng.set_profiling(FAMILIES)
    ag.dynamics_error(params[0], ppm, inits, reals, acts); torch.cuda.synchronize()
    fam = {f: eng.kernel_stats(f) for f in FAMILIES}
    eng.set_profiling([]); eng.reset_stats()
    out["kernel_ms_one_evaluation"] = {f: round(v[0], 4) for f, v in fam.items() if v[1]}
    out["launches_one_evaluation"] = {f: v[1] for f, v in fam.items() if v[1]}
    out["kernel_ms_total_one_evaluation"] = sum(v[0] for v in fam.values())
    return out


CONFIGS = {"rope1": lambda: homogeneous("rope", 1, 1, 10, 20), "cloth1": lambda: homogeneous("cloth", 1, 2, 10, 20),
           "rope64": lambda: homogeneous("rope", 64, 2, 10, 10),
           "granular": lambda: homogeneous("granular", 256, 2, 10, 3), "mixed": lambda: mixed(512, 20, 3), "ppm": ppm_sweep}

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated subset of " + ",".join(CONFIGS))
    args = ap.parse_args()
    for name in (args.only.split(",") if args.only else CONFIGS):
        print(json.dumps(dict(name=name, **CONFIGS[name]())), flush=True)
