mkdir -p gpurun_out/pmc3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_more.py -m gpu -q -k "costs or chamfer" 2>&1 | tail -2
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do tag=$(echo $set | cut -d" " -f1)
  AG_PRECISION=bf16x3 AG_STREAMS=1 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc3 -o $tag -- python3 bench.py --steps 1 --warmup 1 --candidates 128 --no-cpu-baseline --no-kernel-profile --no-bf16x3 > gpurun_out/pmc3/$tag.log 2>&1 || echo FAIL $tag
done
python3 - <<'PY'
import csv, collections
for tag in ['SQ_WAVE_CYCLES','SQ_LDS_BANK_CONFLICT']:
    rows=list(csv.DictReader(open(f'gpurun_out/pmc3/{tag}_counter_collection.csv')))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=set()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'b3' not in k and 'k_chamfer' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp.add((k,r['Dispatch_Id']))
    nd=collections.Counter(k for k,_ in disp)
    for k in agg: print(tag, k, nd[k], {c: f"{v/nd[k]:.4g}" for c,v in agg[k].items()})
kt=list(csv.DictReader(open('gpurun_out/pmc3/SQ_WAVE_CYCLES_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in kt: d[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    if 'ag::' in k: print(k, len(v), round(sum(v)/len(v)/1e3,1),'us')
PY
