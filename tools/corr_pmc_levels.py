"""Per-level FETCH_SIZE / WRITE_SIZE of the volume lookup from the two rocprofv3 --pmc passes of tools/corr_pmc.sh
(one launch per level and timing loop: the kernels are told apart by name and by grid / launch order)."""
import csv, glob, sys, collections
root = sys.argv[1]
out = {}
for tag in ("fetch", "write"):
    f = sorted(glob.glob(f"{root}/cprof_{tag}/*/*counter_collection.csv"))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "corr_index_forward" in r["Kernel_Name"]]
    # corr_bench launches: warm-up 4 levels, then 10 per level, then 10 x 4 levels
    per = collections.defaultdict(list)
    order = []
    for r in rows:
        key = (r["Kernel_Name"].split("(")[0].replace("void droid::", ""), r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"))
        per[key].append(float(r["Counter_Value"]))
        if key not in order: order.append(key)
    out[tag] = (per, order)
per_f, order = out["fetch"]
per_w, _ = out["write"]
print("kernel / grid: launches, KB fetched per launch (FETCH_SIZE x2: 128-byte requests counted as 64), KB written per launch")
for key in order:
    fk = sum(per_f[key]) / len(per_f[key]); wk = sum(per_w.get(key, [0])) / max(1, len(per_w.get(key, [0])))
    print(f"  {key[0][:48]:48s} grid {key[1]:>9s}: {len(per_f[key]):3d} launches  fetch {2 * fk:10.0f} KB  write {wk:10.0f} KB")
