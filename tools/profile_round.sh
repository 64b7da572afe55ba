#!/usr/bin/env bash
# Evidence run of a round on one MI355X (gpurun box): the driver's bench command, rocprofv3 kernel statistics of the same
# command (four streams) and of the single-stream variant, PMC passes (each counter set in its own run, never together with
# another trace domain), HBM traffic per launch, the other BASELINE configs and the shipped planner configuration.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
# Writes gpurun_out/final/; the summaries that are to be judged are then copied into profiles/ (see profiles/README.md).
set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
# single stream, share_first off: every k_edge_enc launch a full 128-candidate one - the launch shape bench.py's roofline pass times
AG_STREAMS=1 AG_SHARE_FIRST=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o s -- python3 $R/bench.py --no-cpu-baseline --no-bf16x3 --no-mpc-iter > $O/prof_single.log 2>&1
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  AG_STREAMS=1 AG_SHARE_FIRST=0 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_$c.log 2>&1
done
AG_STREAMS=1 AG_SHARE_FIRST=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_sq.log 2>&1
cd $R
for k in "k_edge_enc<4>" "k_node_prop<false, false>" "k_node_prop<true, false>"; do
  n=$(echo $k | sed 's/<4>//; s/<false, false>/false/; s/<true, false>/true/')
  python tools/pmc_traffic.py "$k" $O/traffic_$n.json $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > /dev/null
done
python - <<'PY'
import csv, collections, json
O="gpurun_out/final"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq","pmc_GRBM_GUI_ACTIVE","pmc_FETCH_SIZE","pmc_WRITE_SIZE"):
    rows=list(csv.DictReader(open(f"{O}/{d}/p_counter_collection.csv")))
    gmax=collections.defaultdict(int)
    for r in rows:
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        gmax[n]=max(gmax[n],int(r["Grid_Size"]))
    for r in rows:
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "ag::" in n and int(r["Grid_Size"])==gmax[n]:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={n:{c:sum(v)/len(v) for c,v in cs.items()} for n,cs in acc.items()}
json.dump(out,open(f"{O}/pmc_summary.json","w"),indent=1)
for n,cs in out.items(): print(n,{c:round(v) for c,v in cs.items()})
PY
timeout -k 10 600 python tools/bench_planner.py --sorts 1 > $O/planner_configs.jsonl 2> $O/planner_configs.err
timeout -k 10 300 python tools/bench_planner.py --modes chunked --sorts 0 >> $O/planner_configs.jsonl 2>> $O/planner_configs.err
timeout -k 10 300 python tools/probe_planner_phases.py > $O/planner_phases.jsonl 2> $O/planner_phases.err || true
timeout -k 10 300 python tools/probe_loop_host.py rope,granular,cloth > $O/planner_loop_host.jsonl 2> $O/planner_loop_host.err || true
# work-balanced shards of the planner workload: one rank, then two ranks on this one GPU (gloo)
AG_SHARD_CANDIDATES=4000 timeout -k 10 300 python tools/two_rank_planner_shards.py > $O/work_shards_1rank.json 2> $O/work_shards.err || true
AG_SHARD_CANDIDATES=4000 AG_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/two_rank_planner_shards.py > $O/work_shards_2ranks.json 2>> $O/work_shards.err || true
AG_SHARE_PREFIX=0 timeout -k 10 300 python tools/bench_planner.py --modes chunked --sorts 1 > $O/planner_configs_share0.jsonl 2> $O/planner_configs_share0.err
AG_SHARE_PREFIX=0 AG_SHARE_FIRST=0 timeout -k 10 300 python tools/bench_planner.py --modes chunked --sorts 1 > $O/planner_configs_share00.jsonl 2> $O/planner_configs_share00.err
# the RCCL calls with a world of one rank (bench.py AG_BENCH_FORCE_DIST=1): reward hash equal to the plain line's
AG_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-bf16x3 --no-mpc-iter --no-kernel-profile > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err
timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-bf16x3 --no-mpc-iter --no-kernel-profile > $O/bench_plain_short.json 2> $O/bench_plain_short.err
timeout -k 10 120 python tools/trace_mpc_iter.py > $O/small_call_latency.json 2> /dev/null
cd $R
# (SKIP_CONFIGS=1: the per-config evidence is then run as a call of its own - gpurun -- 'bash tools/profile_configs.sh' - the two
# together can exceed one call's time limit)
[ "${SKIP_CONFIGS:-0}" = 1 ] || bash tools/profile_configs.sh > $O/profile_configs.log 2>&1 || tail -20 $O/profile_configs.log
find $O -type f \( -name '*kernel_trace*' -o -name '*_trace.csv' -o -name '*.db' -o -name '*agent_info*' -o -name '*counter_collection*' \) -delete
du -sh $O
cat $O/bench_default.json | cut -c1-400
