"""Per-launch durations and the gaps between consecutive launches from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <kernel_trace.csv> [last N launches=100] -- prints one line per launch of the tail of the trace plus
the sums (kernel time, gap time, span), i.e. how much of a dependent launch chain is work and how much is boundary."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
tk = tg = 0.0
agg = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    dur = (e - s) / 1e3
    if "-q" not in sys.argv:
        print("%-60s grid %8s wg %5s dur %8.2f us gap %7.2f us" % (name, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), dur, gap))
    tk += dur; tg += max(gap, 0.0)
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += dur; a[2] += max(gap, 0.0)
    prev_end = e
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("launches %d  kernel time %.1f us  gaps %.1f us  span %.1f us" % (len(rows), tk, tg, span))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s x%4d  dur avg %7.2f us  gap-before avg %6.2f us  total %8.1f us" % (k, v[0], v[1] / v[0], v[2] / v[0], v[1]))
