"""rocprofv3 --pmc counter_collection.csv files (one per pass) -> per-launch HBM-side bytes of the full-size launches of
one kernel.  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide reads at 64 B, MI355X_MICROARCH.md);
WRITE_SIZE is taken as is.  Both are in KB.  Usage: pmc_traffic.py <kernel substring> <out.json> <csv> [<csv> ...]"""
import csv, json, sys, collections

kern, out = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
for path in sys.argv[3:]:
    rows = [r for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"]]
    if not rows:
        continue
    gmax = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in vals.items()}
n = {k: len(v) for k, v in vals.items()}
fetch_kb, write_kb = res.get("FETCH_SIZE", 0.0), res.get("WRITE_SIZE", 0.0)
json.dump({"kernel": kern, "launches_averaged": n, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
           "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
           "correction": "FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads), WRITE_SIZE exact; separate --pmc passes"},
          open(out, "w"), indent=1)
print(open(out).read())
