"""Shared helpers for the parity tests (fixture loading, golden-step indexing)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def task_of(npz):
    return json.loads(bytes(npz["task_json"]).decode())


def split_edges(npz, prefix):
    cnt = npz[prefix + "n_edges"]
    off = np.concatenate([[0], np.cumsum(cnt)])
    recv, send = npz[prefix + "recv"], npz[prefix + "send"]
    return [(recv[off[b]:off[b + 1]], send[off[b]:off[b + 1]]) for b in range(len(cnt))]


def golden_step_index(repeat):
    """The reference steps the whole batch to max(repeat[:, li]) per look-ahead step.
    Returns base[li] = index of the first recorded forward of look-ahead step li."""
    repeat = np.atleast_2d(repeat)
    mx = repeat.max(0)
    return np.concatenate([[0], np.cumsum(mx)])[:-1]
