"""CPU leg of bench.py: the numpy oracle on a bounded sample of the benchmarked workload.  TEST INFRASTRUCTURE ONLY.

Run as a CHILD process by bench.py's cpu_baseline leg (the bench process has initialised the GPU and must not fork):

    python -m oracle.cpu_baseline <in.npz> <out.npz> <workers> [<blas threads per worker, default 1>]

in.npz : cloud (N_o,3), actions (P,H,4) = P candidates of the timed batch, task_json, pstep, w::<state_dict key> ...
out.npz: state_seqs (P,H,N_o,3) of those candidates (bench.py compares them with the GPU results: parity_check),
         seconds (wall time of the rollout pool), steps (rollout steps executed), workers,
         margin (P,H): smallest edge-selection margin (adaptigraph_oracle.selection_margin) over the forwards of each
         look-ahead step - computed in a second, UNTIMED pass from the positions the timed pass recorded.  A margin
         below 4*adj_thresh*tol marks a step whose graph a position change within the tolerance would alter.

One candidate per worker process, one BLAS thread per worker (so `workers` cores are really used, which a single
multi-threaded numpy call on small 2k x 150 matrices does not manage).  Follows reference
src/planning/forward_dynamics.py:12-205 through oracle/adaptigraph_oracle.py.
"""
import json
import os
import sys
import time


def _roll(job):
    i, W, pstep, cloud, action, task = job
    from oracle import adaptigraph_oracle as O
    tr = []
    seq = O.dynamics(W, pstep, cloud, action[None], task, trace=tr)["state_seqs"][0]
    return i, seq, [rec["state_last"] for rec in tr[0]]


def _margins(job):
    i, states, rep, task, N_o = job
    import numpy as np
    from oracle import adaptigraph_oracle as O
    N = states[0].shape[0]
    mask = np.ones(N, bool)
    tool = np.zeros(N, bool)
    tool[N_o:] = True
    per = [O.selection_margin(s, task["adj_thresh"], mask, tool, task["topk"]) for s in states]
    out, k = [], 0
    for r in rep:
        out.append(min(per[k:k + int(r)]) if r > 0 else float("inf"))
        k += int(r)
    return i, out


def main(argv):
    src, dst, workers = argv[1], argv[2], int(argv[3])
    blas = str(max(1, int(argv[4]))) if len(argv) > 4 else "1"
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = blas                                 # before numpy loads its BLAS, inherited by the workers
    import numpy as np
    import multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import adaptigraph_oracle as O
    z = np.load(src)
    task = json.loads(bytes(z["task_json"]).decode())
    W = O.weights_from_npz(z)
    cloud, actions, pstep = z["cloud"], z["actions"], int(z["pstep"])
    _, rep = O.decode_action(actions, task["push_length"])
    jobs = [(i, W, pstep, cloud, actions[i], task) for i in range(actions.shape[0])]
    out = np.zeros((actions.shape[0], actions.shape[1], cloud.shape[0], 3), np.float32)
    states = [None] * len(jobs)
    margin = np.zeros(rep.shape, np.float64)
    with mp.get_context("fork").Pool(workers) as pool:
        t0 = time.time()
        for i, seq, st in pool.imap_unordered(_roll, jobs):
            out[i], states[i] = seq, st
        dt = time.time() - t0                                # the timed sample ends here
        mjobs = [(i, states[i], rep[i], task, cloud.shape[0]) for i in range(len(jobs))]
        for i, m in pool.imap_unordered(_margins, mjobs):
            margin[i] = m
    np.savez(dst, state_seqs=out, seconds=dt, steps=int(rep.sum()), workers=workers, margin=margin)


if __name__ == "__main__":
    main(sys.argv)
