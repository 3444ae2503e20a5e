"""GPU busy time of the training step from a rocprofv3 --kernel-trace CSV: union of the kernel intervals, idle gaps by size, and the
time during which 1 / 2+ kernels were resident.  Usage: python tools/gpu_busy.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# keep the last `steps` steps: find the step period from the fused Adam launches (7 per step: last one closes the step)
adam = [e for e in ev if "fused_adam" in e[2].lower() or "FusedAdam" in e[2]]
per = len(adam) // max(1, (len(adam) // 7))
marks = [adam[i][1] for i in range(6, len(adam), 7)]
if len(marks) < steps + 1:
    print("not enough steps in the trace:", len(marks)); sys.exit(1)
t0, t1 = marks[-steps - 1], marks[-1]
win = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
pts = []
for s, e, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy1 = busy2 = 0; depth = 0; last = t0; gaps = []
for t, d in pts:
    if depth == 0 and t > last: gaps.append(t - last)
    if depth == 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
if t1 > last: gaps.append(t1 - last)
tot = (t1 - t0) / steps / 1e6
print("step %.2f ms | one kernel resident %.2f ms | two or more %.2f ms | idle %.2f ms (%d gaps/step)" %
      (tot, busy1 / steps / 1e6, busy2 / steps / 1e6, sum(gaps) / steps / 1e6, len(gaps) // steps))
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e9)):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps %6.0f - %6.0f us: %5d per step, %.3f ms per step" % (lo / 1e3, hi / 1e3, len(g) // steps, sum(g) / steps / 1e6))
print("sum of kernel durations per step %.2f ms over %d launches" % (sum(e - s for s, e, _ in win) / steps / 1e6, len(win) // steps))
# the largest idle gaps and what ran before / after them
win.sort()
cur_end = t0; big = []
for s, e, n in win:
    if s > cur_end: big.append((s - cur_end, n))
    cur_end = max(cur_end, e)
big.sort(reverse=True)
print("largest gaps (us) and the kernel that ended them:")
for g, n in big[:15]:
    print("  %8.1f  %s" % (g / 1e3, n[:90]))
