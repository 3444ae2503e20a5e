"""GPU busy vs wall time of the training step from a rocprofv3 --kernel-trace CSV: union of all kernel intervals (any stream)
over the span of the last N steps.  Usage: python tools/gpu_idle.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# keep the last 60 % of the trace (warm-up excluded)
t_lo = iv[0][0] + int(0.4 * (iv[-1][1] - iv[0][0]))
iv = [x for x in iv if x[0] >= t_lo]
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps.sort(reverse=True)
print("span %.1f ms, GPU busy (union over streams) %.1f ms = %.1f %%, idle %.1f ms in %d gaps; gaps > 20 us: %d (%.1f ms), 5-20 us: %d (%.1f ms), < 5 us: %d (%.1f ms)" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps),
    sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6,
    sum(1 for g in gaps if 5000 < g <= 20000), sum(g for g in gaps if 5000 < g <= 20000) / 1e6,
    sum(1 for g in gaps if g <= 5000), sum(g for g in gaps if g <= 5000) / 1e6))
print("kernels in window:", len(iv), " sum of kernel durations %.1f ms" % (sum(e - s for s, e in iv) / 1e6))
