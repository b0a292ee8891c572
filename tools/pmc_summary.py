"""Summarises rocprofv3 --pmc counter_collection.csv files into per-kernel averages (JSON)."""
import collections, csv, glob, json, sys
out = collections.defaultdict(dict)
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "droid::" not in n:
                continue
            key = n.split("(")[0].replace("void ", "")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                out[k][c] = dict(mean=sum(x) / len(x), launches=len(x))
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(k, {c: round(x["mean"], 1) for c, x in v.items()})
