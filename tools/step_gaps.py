"""Reduce a rocprofv3 --kernel-trace of bench.py: what happens on the GPU between the last rollout kernel of one timed step and
the first chain kernel of the next (cost kernels, collectives, the next call's plan / init / graph kernels), in microseconds.
  python tools/step_gaps.py DIR"""
import csv, glob, os, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
plans = [i for i, r in enumerate(rows) if "k_roll_plan" in r["Kernel_Name"]]
for pi in plans[3:6]:
    t_plan = int(rows[pi]["Start_Timestamp"])
    # last chain kernel before the plan kernel, first chain kernel after it
    prev = max(i for i in range(pi) if "k_node_prop" in rows[i]["Kernel_Name"] or "k_edge_enc" in rows[i]["Kernel_Name"])
    nxt = min(i for i in range(pi, len(rows)) if "k_edge_enc" in rows[i]["Kernel_Name"] or "k_node_prop" in rows[i]["Kernel_Name"])
    t0 = max(int(rows[i]["End_Timestamp"]) for i in range(max(0, prev - 8), prev + 1))
    print(f"--- gap between chain kernels: {(int(rows[nxt]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for i in range(prev + 1, nxt + 1):
        r = rows[i]
        print(f"   +{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  "
              f"q{r.get('Queue_Id', '?')}  {name(r)[:70]}")
