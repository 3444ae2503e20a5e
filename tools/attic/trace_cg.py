import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/pc/**/*kernel_trace.csv", recursive=True)[0])))
sel = [r for r in rows if "cg_" in r["Kernel_Name"]]
# the script runs 7 configs x (1 warm + 5 timed) grid calls x 2 directions; print per config the mean per kernel over the timed calls
names = []
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("sc::", "")
    if n not in names: names.append(n)
per_call = collections.Counter(r["Kernel_Name"].split("(")[0].replace("sc::", "") for r in sel)
calls = per_call["cg_meta_kernel"]
print("direction passes:", calls)
cfg = calls // 7
by = collections.defaultdict(list)
idx = collections.Counter()
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("sc::", "")
    k = idx[n]; idx[n] += 1
    per_pass = per_call[n] // calls
    c = (k // per_pass) // cfg
    by[(c, n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for c in range(7):
    print("config", c, " ".join("%s %.0f" % (n.replace("cg_", "").replace("_kernel", ""), sum(by[(c, n)]) / max(1, len(by[(c, n)])) * (per_call[n] // calls)) for n in names))
