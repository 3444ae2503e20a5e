"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV (name shortened)."""
import csv, sys, re
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:top]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:70]
    print("%-70s calls %5s total %9.3f ms avg %9.1f us  %5.1f%%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
