"""From a rocprofv3 kernel trace (csv) of a two-stream run: for every kernel family, its mean duration when its whole
lifetime lay inside kernels of the OTHER queue that are MFMA-bound (F) or memory-bound (M), or alone.  Diagnostic."""
import csv, sys, collections, bisect
MFMA = ("k_edge_enc", "k_node_prop", "k_node_enc")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ag::" not in n: continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "").replace("ag::", ""), r["Queue_Id"]))
rows.sort()
byq = collections.defaultdict(list)
for s, e, n, q in rows: byq[q].append((s, e, n))
qs = list(byq)
acc = collections.defaultdict(list)
for q in qs:
    others = [x for oq in qs if oq != q for x in byq[oq]]
    others.sort()
    starts = [x[0] for x in others]
    for s, e, n in byq[q]:
        i = bisect.bisect_right(starts, s) - 1
        cover = None
        if i >= 0 and others[i][1] >= e:            # fully inside one kernel of the other queue
            cover = "F" if any(m in others[i][2] for m in MFMA) else "M"
        elif i >= 0 and others[i][1] > s or (i + 1 < len(others) and others[i + 1][0] < e):
            cover = "mixed"
        else:
            cover = "alone"
        acc[(n, cover)].append(e - s)
for (n, c), v in sorted(acc.items()):
    if len(v) >= 5: print(f"{n:24s} {c:6s} n={len(v):5d} mean {sum(v)/len(v)/1e3:9.1f} us")
