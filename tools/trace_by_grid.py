"""Average GPU duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV.  Usage: python tools/trace_by_grid.py <kernel_trace.csv> [filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name: continue
    short = name.split("(")[0][:60]
    key = (short, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, [0, 0.0, 1e9])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for (n, g, w), (c, t, m) in acc.items():
    print("%-60s grid %8s wg %5s calls %4d avg %7.1f us min %7.1f" % (n, g, w, c, t / c, m))
