"""Overlap report from a rocprofv3 kernel trace (csv): how much wall time the rollout spends with an MFMA-bound
kernel alone, a memory-bound kernel alone, MFMA || memory, MFMA || MFMA, memory || memory.  Diagnostic only."""
import csv, sys, collections

MFMA = ("k_edge_enc", "k_node_prop", "k_node_enc")
def kind(name):
    return "F" if any(m in name for m in MFMA) else "M"

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "ag::" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
# restrict to the window given on the command line as fractions of the trace (default: everything)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
ev = []
for s, e, n, q in rows:
    if e <= a or s >= b: continue
    ev.append((max(s, a), 1, kind(n))); ev.append((min(e, b), -1, kind(n)))
ev.sort()
cnt = {"F": 0, "M": 0}
acc = collections.Counter()
prev = ev[0][0]
for t, d, k in ev:
    key = "F%d/M%d" % (min(cnt["F"], 2), min(cnt["M"], 2))
    acc[key] += t - prev
    prev = t
    cnt[k] += d
tot = sum(acc.values())
print("window %.1f ms" % (tot / 1e6))
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print("  %-6s %8.1f ms  %5.1f%%" % (k, v / 1e6, 100.0 * v / tot))
# per-kernel average duration inside the window
d = collections.defaultdict(list)
for s, e, n, q in rows:
    if s >= a and e <= b: d[n].append(e - s)
for n, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    print("  %-28s n=%5d avg %8.1f us  total %8.1f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
print("queues:", collections.Counter(r[3] for r in rows))
