"""bench.py's `parity_check.vs_reference` compares candidates of the TIMED batch with outputs of the reference itself
(tests/golden/full_cloth_{a,flip}.npz, written by tests/golden/make_golden.py --fullsize from the imported reference's dynamics()).
That is only meaningful if the fixtures' inputs ARE the bench's: checked here on the CPU, bit for bit - start state, weights, task
scalars and the raw actions of candidates 0, 49, 487, 1023 - and a fixture whose inputs differ in one bit must not be matched."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench_inputs():
    import bench as B
    rng = np.random.default_rng(0)                           # bench.py: main()
    cloud = B.cloth_cloud(45, rng)
    task = B.make_task(max_nR=int(1.2 * 6 * (cloud.shape[0] + 1)) + 64)
    W = B.random_weights(0)
    actions = B.make_actions(1024, 2, 10, cloud, rng)
    return B, cloud, task, W, actions


def test_reference_fixtures_are_candidates_of_the_timed_batch():
    B, cloud, task, W, actions = _bench_inputs()
    ref = B.reference_golden(cloud, task, W, actions)
    assert sorted(ref) == [0, 49, 487, 1023]
    for c, seq in ref.items():
        assert seq.shape == (2, cloud.shape[0], 3) and seq.dtype == np.float32 and np.isfinite(seq).all()
        assert float(np.abs(seq[0] - cloud).max()) > 1e-3     # a rollout, not the start state


def test_a_fixture_with_other_inputs_is_not_matched():
    B, cloud, task, W, actions = _bench_inputs()
    a2 = actions.copy()
    a2[49, 0, 2] = np.nextafter(a2[49, 0, 2], np.float32(10))   # one ulp in one action
    assert sorted(B.reference_golden(cloud, task, W, a2)) == [0, 487, 1023]
    c2 = cloud.copy()
    c2[7, 1] = np.nextafter(c2[7, 1], np.float32(1))
    assert B.reference_golden(c2, task, W, actions) == {}
    W2 = dict(W)
    k = "relation_encoder.model.2.weight"
    W2[k] = W[k].copy()
    W2[k][3, 5] = np.nextafter(W2[k][3, 5], np.float32(1))
    assert B.reference_golden(cloud, task, W2, actions) == {}
    assert B.reference_golden(cloud, dict(task, topk=6), W, actions) == {}
