"""The committed PMC traffic summaries are inputs of bench.py (roofline.traffic): keep their shape checked on CPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_summaries_carry_what_bench_reads():
    want = {"r03_traffic_k_edge_enc.json": "edges_per_launch", "r03_traffic_k_node_prop.json": "candidates_per_launch",
            "r03_traffic_k_node_final.json": "candidates_per_launch"}
    for name, key in want.items():
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["hbm_bytes_per_launch"] > 0 and d[key] > 0, name
        assert "FETCH_SIZE x2" in d["correction"]


def test_bench_default_profile_is_a_bench_line():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["roofline"]["traffic"] and d["parity_check"]["ok"]
    # the names bench.py looks up must be the files that are committed
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("r03_traffic_k_edge_enc.json", "r03_traffic_k_node_prop.json"):
        assert name in src and os.path.exists(os.path.join(ROOT, "profiles", name)), name
