#!/usr/bin/env bash
# Second part of a round's evidence run (r06 additions), a gpurun call of its own after tools/profile_round.sh (the two together exceed
# one call's time limit): writes into the same gpurun_out/final/ without clearing it.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round_extra.sh'
set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
# random_interact.py's planner configuration (n_update_iter 5, n_sample 1000 as 1 / 2 chunks), strict vs default
timeout -k 10 300 python tools/bench_interact.py > $O/interact_configs.jsonl 2> $O/interact_configs.err || true
# the reference's unchanged 40-call loop with planner_config['group']: one rank, then two ranks on this one GPU (gloo)
AG_LOOP_CHUNKS=40 timeout -k 10 300 python tools/two_rank_planner_loop.py > $O/planner_loop_1rank.json 2> $O/planner_loop_ranks.err || true
AG_LOOP_CHUNKS=40 AG_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 tools/two_rank_planner_loop.py > $O/planner_loop_2ranks.json 2>> $O/planner_loop_ranks.err || true
# `python bench.py --gpus 2` with NO launcher in front, full size: the parent starts its own two ranks (rehearsal: both on this GPU, gloo)
AG_BENCH_SHARE_GPU=1 AG_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-bf16x3 --no-mpc-iter --no-kernel-profile > $O/bench_bare_two_ranks.json 2> $O/bench_bare_two_ranks.err || true
# HIP-API + kernel traces of the interact configuration (blocking HIP calls per planner call), rope
export TMPDIR=/tmp
cd /tmp
for m in interact1 interact1_strict interact2 interact2_strict; do
  AG_TRACE_META_DIR=$O/tr_rope_$m timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/tr_rope_$m -o t -- python3 $R/tools/trace_planner_loop.py run rope $m > $O/tr_rope_$m.log 2>&1 || true
  python3 $R/tools/trace_planner_loop.py report $O/tr_rope_$m > $O/planner_trace_rope_$m.json || true
done
cd $R
# secondary bf16x3 mode, A/B of ONE 8-wavefront workgroup per CU sharing one weight ring (tools/build_experiment.sh b3wg512 -DAG_B3_WG512)
# against the shipped two 4-wavefront workgroups: bench line of each, then PMC passes of each (own runs, no other trace domain)
if [ -f $R/adaptigraph_amd/csrc/libadaptigraph_hip_b3wg512.so ]; then
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mpc-iter > $O/b3_base.json 2> $O/b3_base.err || true
  ADAPTIGRAPH_AMD_LIB=$R/adaptigraph_amd/csrc/libadaptigraph_hip_b3wg512.so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mpc-iter > $O/b3_wg512.json 2> $O/b3_wg512.err || true
  cd /tmp
  for v in base wg512; do
    lib=$R/adaptigraph_amd/csrc/libadaptigraph_hip.so; [ $v = wg512 ] && lib=$R/adaptigraph_amd/csrc/libadaptigraph_hip_b3wg512.so
    ADAPTIGRAPH_AMD_LIB=$lib AG_STREAMS=1 AG_SHARE_FIRST=0 AG_PRECISION=bf16x3 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_b3_${v}_sq -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_b3_${v}_sq.log 2>&1 || true
    ADAPTIGRAPH_AMD_LIB=$lib AG_STREAMS=1 AG_SHARE_FIRST=0 AG_PRECISION=bf16x3 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_b3_${v}_lds -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_b3_${v}_lds.log 2>&1 || true
  done
  cd $R
  python - <<'PY'
import csv, collections, json, os
O="gpurun_out/final"
out={}
for v in ("base","wg512"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (f"pmc_b3_{v}_sq", f"pmc_b3_{v}_lds"):
        p=f"{O}/{d}/p_counter_collection.csv"
        if not os.path.exists(p): continue
        rows=list(csv.DictReader(open(p)))
        gmax=collections.defaultdict(int)
        for r in rows:
            n=r["Kernel_Name"].split("(")[0].replace("void ","")
            gmax[n]=max(gmax[n],int(r["Grid_Size"]))
        for r in rows:
            n=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "_b3" in n and int(r["Grid_Size"])==gmax[n]:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[v]={n:{c:sum(x)/len(x) for c,x in cs.items()} for n,cs in acc.items()}
    for n,cs in out[v].items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            cs["mfma_busy_frac"]=cs["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(cs["GRBM_GUI_ACTIVE"]/8)
json.dump(out,open(f"{O}/pmc_b3_ab_summary.json","w"),indent=1)
print(json.dumps(out)[:1500])
PY
fi
find $O -type f \( -name '*kernel_trace*' -o -name '*_trace.csv' -o -name '*.db' -o -name '*agent_info*' -o -name '*counter_collection*' \) -delete
du -sh $O
