"""The committed PMC traffic summaries are inputs of bench.py (roofline.traffic): keep their shape checked on CPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = ("r06", "r05", "r04", "r03")                                       # bench.py takes the newest round's file of a name


def _latest(stem):
    for r in ROUNDS:
        p = os.path.join(ROOT, "profiles", f"{r}_{stem}")
        if os.path.exists(p):
            return p
    raise AssertionError(f"no committed profiles/*_{stem}")


def test_traffic_summaries_carry_what_bench_reads():
    want = {"traffic_k_edge_enc.json": "edges_per_launch", "traffic_k_node_prop.json": "candidates_per_launch",
            "traffic_k_node_final.json": "candidates_per_launch"}
    for stem, key in want.items():
        d = json.load(open(_latest(stem)))
        assert d["hbm_bytes_per_launch"] > 0 and d[key] > 0, stem
        assert "FETCH_SIZE x2" in d["correction"]


def test_bench_default_profile_is_a_bench_line():
    d = json.load(open(_latest("bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["roofline"]["traffic"] and d["parity_check"]["ok"]
    if "candidate_forwards_executed" in d["end_to_end"]:          # r05 accounting: recomputable from the line alone
        e = d["end_to_end"]
        flop = e["flop_per_step_executed"] * e["candidate_forwards_executed"] - e["edges_not_encoded_thanks_to_the_shared_first_forward"] * d["roofline"]["flop_per_edge"]
        assert abs(flop - e["flop_per_call_executed"]) <= 1e-6 * flop
        assert abs(e["executed_tflops"] - flop / (d["ms_per_step"] * 1e-3) / 1e12) <= 1e-6 * e["executed_tflops"]
        assert e["candidate_forwards_needed"] == d["config"]["candidates"] * d["config"]["horizon"]
        v = d["parity_check"]["vs_reference"]
        assert v["n_candidates"] >= 60 and v["ok"] and all(f["near_tie_in_the_reference"] for f in v["flips_vs_reference"])
        assert all(f["checker_induced"] is not None for f in d["parity_check"]["edge_flips"])
        if "ok_clause" in v:                                     # r06: the line says which clause held, flips carry their reach
            assert v["ok_clause"].startswith(("every covered candidate within tol", f"{v['candidates_within_tol_all_steps']} of {v['n_candidates']} within tol"))
            first = {}
            for f in v["flips_vs_reference"]:                    # a candidate's FIRST step out of tolerance is the attributed one
                if f["candidate"] not in first or f["lookahead_step"] < first[f["candidate"]]["lookahead_step"]:
                    first[f["candidate"]] = f
            assert all(f["reference_selection_margin"] < f["margin_within_reach_of_the_deviation_before"] for f in first.values())
            assert d["env"]["hw_queues"]["GPU_MAX_HW_QUEUES"] == d["env"]["GPU_MAX_HW_QUEUES"]
    assert len(d["reward_sha256"]) == 64        # bench.py compares every default-workload line's reward SHA with this one (any N)
    # the names bench.py looks up must be the files that are committed
    src = open(os.path.join(ROOT, "bench.py")).read()
    for stem in ("traffic_k_edge_enc.json", "traffic_k_node_prop.json"):
        assert stem in src and all(f'"{r}"' in src for r in ROUNDS), stem
