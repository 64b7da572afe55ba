"""CPU leg of bench.py: the numpy oracle on a bounded sample of the benchmarked workload.  TEST INFRASTRUCTURE ONLY.

Run as a CHILD process by bench.py's cpu_baseline leg (the bench process has initialised the GPU and must not fork):

    python -m oracle.cpu_baseline <in.npz> <out.npz> <workers>

in.npz : cloud (N_o,3), actions (P,H,4) = P candidates of the timed batch, task_json, pstep, w::<state_dict key> ...
out.npz: state_seqs (P,H,N_o,3) of those candidates (bench.py compares them with the GPU results: parity_check),
         seconds (wall time of the pool), steps (rollout steps executed), workers.

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
    return i, O.dynamics(W, pstep, cloud, action[None], task)["state_seqs"][0]


def main(argv):
    src, dst, workers = argv[1], argv[2], int(argv[3])
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"                                  # before numpy loads its BLAS, inherited by the workers
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
    t0 = time.time()
    with mp.get_context("fork").Pool(workers) as pool:
        for i, seq in pool.imap_unordered(_roll, jobs):
            out[i] = seq
    dt = time.time() - t0
    np.savez(dst, state_seqs=out, seconds=dt, steps=int(rep.sum()), workers=workers)


if __name__ == "__main__":
    main(sys.argv)
