#!/usr/bin/env bash
# Per-kernel evidence for the BASELINE configs that are not the bench line (configs[1], [2], [4]) and for the secondary
# bf16x3 arithmetic: rocprofv3 kernel statistics on ONE in-library stream (durations then belong to the kernel alone), the
# per-config report of tools/bench_configs.py (HIP events: dominant kernel, algorithmic FLOP rate against the fp32 MFMA
# peak, edge builder share), and PMC passes (own runs, no other trace domain) for granular and bf16x3.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_configs.sh'
# Writes gpurun_out/cfg/; summaries to be judged are copied into profiles/ (profiles/README.md).
set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cfg
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other_configs.err
cd /tmp
for c in rope64 granular mixed; do   # (sharings off: one launch shape per kernel, like the HIP-event report)
  AG_STREAMS=1 AG_SHARE_FIRST=0 AG_SHARE_PREFIX=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o s -- python3 $R/tools/bench_configs.py --only $c > $O/prof_$c.log 2>&1
done
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_PRECISION=bf16x3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3 -o s -- python3 $R/bench.py --no-cpu-baseline --no-bf16x3 --no-mpc-iter --steps 2 --warmup 1 > $O/prof_b3.log 2>&1
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_PRECISION=bf16x3 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_b3_sq -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_b3_sq.log 2>&1
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_PRECISION=bf16x3 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b3_grbm -o p -- python3 $R/bench.py --candidates 256 --steps 1 --warmup 0 --no-cpu-baseline --no-bf16x3 --no-kernel-profile --no-mpc-iter > $O/pmc_b3_grbm.log 2>&1
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_SHARE_PREFIX=0 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_gran_sq -o p -- python3 $R/tools/bench_configs.py --only granular > $O/pmc_gran_sq.log 2>&1
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_SHARE_PREFIX=0 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gran_grbm -o p -- python3 $R/tools/bench_configs.py --only granular > $O/pmc_gran_grbm.log 2>&1
AG_STREAMS=1 AG_SHARE_FIRST=0 AG_SHARE_PREFIX=0 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_gran_valu -o p -- python3 $R/tools/bench_configs.py --only granular > $O/pmc_gran_valu.log 2>&1
cd $R
python - <<'PY'
import csv, collections, json, os
O="gpurun_out/cfg"
for tag, dirs in (("b3", ("pmc_b3_sq","pmc_b3_grbm")), ("gran", ("pmc_gran_sq","pmc_gran_grbm","pmc_gran_valu"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        p=f"{O}/{d}/p_counter_collection.csv"
        if not os.path.exists(p): continue
        rows=list(csv.DictReader(open(p)))
        gmax=collections.defaultdict(int)
        for r in rows:
            n=r["Kernel_Name"].split("(")[0].replace("void ","")
            gmax[n]=max(gmax[n],int(r["Grid_Size"]))
        for r in rows:
            n=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "ag::" in n and int(r["Grid_Size"])==gmax[n]:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out={n:{c:sum(v)/len(v) for c,v in cs.items()} for n,cs in acc.items()}
    for n,cs in out.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            cs["mfma_busy_frac"]=cs["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(cs["GRBM_GUI_ACTIVE"]/8)
    json.dump(out,open(f"{O}/pmc_{tag}_summary.json","w"),indent=1)
    for n,cs in out.items(): print(tag,n,{c:round(v,3) for c,v in cs.items()})
PY
find $O -type f \( -name '*kernel_trace*' -o -name '*.db' -o -name '*agent_info*' -o -name '*counter_collection*' \) -delete
du -sh $O
cat $O/other_configs.jsonl | cut -c1-600
