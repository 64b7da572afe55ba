"""Copy the summaries of an evidence run (tools/profile_round.sh -> gpurun_out/final, gpurun_out/cfg) into profiles/<round>_*.
Only reductions are copied (json / jsonl / the rocprofv3 --stats csv); traces and counter dumps stay in gpurun_out/.
  python tools/collect_profiles.py r04"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
F, C, P = (os.path.join(ROOT, "gpurun_out", d) for d in ("final", "cfg", "")); P = os.path.join(ROOT, "profiles")


def bench_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][0])


def stats(src, dst):
    """rocprofv3 --stats kernel table, our kernels and the few torch ones that matter (rows below 0.005 % dropped)"""
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if float(r["Percentage"]) >= 0.005 or "ag::" in r["Name"]]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        for r in keep:
            r["Name"] = r["Name"][:160]
            w.writerow(r)


line = bench_line(os.path.join(F, "bench_default.json"))
json.dump(line, open(os.path.join(P, f"{rnd}_bench_default.json"), "w"), indent=1)
stats(os.path.join(F, "prof_default", "d_kernel_stats.csv"), os.path.join(P, f"{rnd}_bench_default_kernel_stats.csv"))
stats(os.path.join(F, "prof_single", "s_kernel_stats.csv"), os.path.join(P, f"{rnd}_bench_single_stream_kernel_stats.csv"))
E_enc, cand = line["config"]["edges_encoded_per_graph"], 128
src = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) --kernel-trace -- python3 bench.py --candidates 256 --steps 1 --warmup 0 "
       "--no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter, AG_STREAMS=1 AG_SHARE_FIRST=0 (tools/profile_round.sh); "
       "tools/pmc_traffic.py")
for src_name, dst_name in (("traffic_k_edge_enc.json", "traffic_k_edge_enc.json"), ("traffic_k_node_propfalse.json", "traffic_k_node_prop.json"),
                           ("traffic_k_node_proptrue.json", "traffic_k_node_final.json")):
    d = json.load(open(os.path.join(F, src_name)))
    d.update(candidates_per_launch=cand, edges_per_launch=E_enc * cand, source=src)
    json.dump(d, open(os.path.join(P, f"{rnd}_{dst_name}"), "w"), indent=1)
pmc = json.load(open(os.path.join(F, "pmc_summary.json")))
for n, c in pmc.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        c["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
json.dump(pmc, open(os.path.join(P, f"{rnd}_pmc_counters.json"), "w"), indent=1)
shutil.copy(os.path.join(F, "planner_configs.jsonl"), os.path.join(P, f"{rnd}_planner_configs.jsonl"))
with open(os.path.join(P, f"{rnd}_planner_configs.jsonl"), "a") as f:
    for name, tag in (("planner_configs_share0.jsonl", {"share_prefix": 0}), ("planner_configs_share00.jsonl", {"share_prefix": 0, "share_first": 0})):
        if os.path.exists(os.path.join(F, name)):
            for l in open(os.path.join(F, name)):
                r = json.loads(l); r.update(tag)
                f.write(json.dumps(r) + "\n")
one, plain = bench_line(os.path.join(F, "bench_one_rank_rccl.json")), bench_line(os.path.join(F, "bench_plain_short.json"))
json.dump({"what": "bench.py --gpus 1 --steps 5 --warmup 2 with AG_BENCH_FORCE_DIST=1: nccl (= RCCL) process group with a world of one rank, "
                   "the all-gather of the rewards and both MAX all-reduces issued on it; against the plain run on the same box",
           "one_rank_rccl": {k: one[k] for k in ("value", "ms_per_step", "reward_sha256", "multi_gpu")},
           "plain": {k: plain[k] for k in ("value", "ms_per_step", "reward_sha256")},
           "reward_vectors_bit_equal": one["reward_sha256"] == plain["reward_sha256"]},
          open(os.path.join(P, f"{rnd}_one_rank_rccl.json"), "w"), indent=1)
shutil.copy(os.path.join(F, "small_call_latency.json"), os.path.join(P, f"{rnd}_small_call_latency.json"))
for name in ("planner_phases.jsonl", "planner_loop_host.jsonl"):
    if os.path.exists(os.path.join(F, name)) and os.path.getsize(os.path.join(F, name)):
        shutil.copy(os.path.join(F, name), os.path.join(P, f"{rnd}_{name}"))
ws = {}
for tag in ("1rank", "2ranks"):
    pth = os.path.join(F, f"work_shards_{tag}.json")
    rows = [l for l in open(pth) if l.startswith("{")] if os.path.exists(pth) else []
    if rows:
        ws[tag] = json.loads(rows[-1])
if ws:
    json.dump({"what": "tools/two_rank_planner_shards.py: 4000 pushes of the shipped rope sampler sharded by WORK (adaptigraph_amd.rollout_work) "
                       "and by count, one rank and two ranks on one GPU (gloo); reward hashes must agree", **ws},
              open(os.path.join(P, f"{rnd}_work_shards.json"), "w"), indent=1)
# r06: the interact configuration, the rank-dealt planner loop, the bare two-rank launch, blocking-call traces
def _last_json(path):
    rows = [l for l in open(path) if l.startswith("{")] if os.path.exists(path) else []
    return json.loads(rows[-1]) if rows else None


if os.path.exists(os.path.join(F, "interact_configs.jsonl")) and os.path.getsize(os.path.join(F, "interact_configs.jsonl")):
    shutil.copy(os.path.join(F, "interact_configs.jsonl"), os.path.join(P, f"{rnd}_interact_configs.jsonl"))
ranks = {tag: _last_json(os.path.join(F, f"planner_loop_{tag}.json")) for tag in ("1rank", "2ranks")}
if all(ranks.values()):
    json.dump({"what": "tools/two_rank_planner_loop.py: the reference's UNCHANGED 40-call planner loop (plan.py:210, 241-247) on the shipped rope "
                       "configuration with planner_config['group'], one rank and two ranks on ONE GPU (gloo): call ci on rank ci % world, "
                       "merge_res all-gathers the winners; result and generator SHA-256 must not depend on the number of ranks (two ranks "
                       "share the card here, so the time says nothing about scaling)", **ranks,
               "result_bit_equal": ranks["1rank"]["result_sha256"] == ranks["2ranks"]["result_sha256"]},
              open(os.path.join(P, f"{rnd}_planner_loop_ranks.json"), "w"), indent=1)
bare = _last_json(os.path.join(F, "bench_bare_two_ranks.json"))
if bare:
    json.dump({"what": "AG_BENCH_SHARE_GPU=1 AG_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 ... with NO launcher in front, full "
                       "size: the parent starts python -m torch.distributed.run itself (multi_gpu.launched_by); both ranks share the one GPU of "
                       "the box, so ms_per_step is not a scaling figure", "line": {k: bare[k] for k in ("value", "n_gpus", "ms_per_step", "reward_sha256", "multi_gpu", "config")},
               "one_rank_reward_sha256": plain["reward_sha256"], "reward_vectors_bit_equal": bare["reward_sha256"] == plain["reward_sha256"]},
              open(os.path.join(P, f"{rnd}_bare_two_rank_launch.json"), "w"), indent=1)
for m in ("interact1", "interact1_strict", "interact2", "interact2_strict"):
    src_t = os.path.join(F, f"planner_trace_rope_{m}.json")
    if os.path.exists(src_t) and os.path.getsize(src_t):
        shutil.copy(src_t, os.path.join(P, f"{rnd}_planner_trace_rope_{m}.json"))
# per-config evidence (tools/profile_configs.sh)
shutil.copy(os.path.join(C, "other_configs.jsonl"), os.path.join(P, f"{rnd}_other_configs.jsonl"))
for c in ("rope64", "granular", "mixed"):
    stats(os.path.join(C, f"prof_{c}", "s_kernel_stats.csv"), os.path.join(P, f"{rnd}_{c}_kernel_stats.csv"))
stats(os.path.join(C, "prof_b3", "s_kernel_stats.csv"), os.path.join(P, f"{rnd}_bf16x3_kernel_stats.csv"))
for tag, name in (("b3", "bf16x3_pmc_counters.json"), ("gran", "granular_pmc_counters.json")):
    shutil.copy(os.path.join(C, f"pmc_{tag}_summary.json"), os.path.join(P, f"{rnd}_{name}"))
print("collected into profiles/", rnd)
