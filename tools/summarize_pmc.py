"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (largest dispatch of each hand-written kernel)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if not name.startswith("sc::"):
        continue
    key = (name, r.get("Grid_Size", r.get("Grid_Size_X")))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
for key in sorted(agg):
    vals = {c: v / cnt[(key, c)] for c, v in agg[key].items()}
    print(key[0], "grid", key[1], " ".join("%s=%.4g" % (c, v) for c, v in sorted(vals.items())))
