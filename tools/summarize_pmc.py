"""Per-kernel summary of a rocprofv3 --pmc counter_collection CSV: largest dispatch (by first counter) of each sc:: kernel."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(dict)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if "sc::" not in name:
        continue
    d[(r["Dispatch_Id"], name)][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (did, name), c in d.items():
    k = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in c else sorted(c)[0]
    if name not in best or c.get(k, 0) > best[name].get(k, 0):
        best[name] = c
for name, c in sorted(best.items()):
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print("%-60s wave_cyc %.3g wait_any %2.0f%% wait_inst %2.0f%% active %2.0f%% mfma_busy_cyc %.3g busy_cyc %.3g lds_conf %.1f%%" % (
            name[-60:], wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
    else:
        print("%-60s %s" % (name[-60:], " ".join("%s=%.4g GB" % (k, v / 1e6) for k, v in sorted(c.items()))))
