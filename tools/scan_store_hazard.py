"""Scan the compiler's output for the gfx950 store-data hazard LLVM does not cover (tools/micro/store_war_hazard.hip): a
buffer_store_dwordx3 / x4 whose soffset is an SGPR, followed IMMEDIATELY by a VALU instruction that writes one of its data registers.
    for f in csrc/*.hip: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only f -o x.s;  python tools/scan_store_hazard.py x.s ..."""
import re
import sys

store = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(\S+)")
dst = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
total = hits = 0
for path in sys.argv[1:]:
    kernel = "?"
    lines = open(path).read().splitlines()
    ins = []
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        ins.append((kernel, s))
    for i, (k, s) in enumerate(ins[:-1]):
        m = store.match(s)
        if not m:
            continue
        lo, hi, soff = int(m.group(1)), int(m.group(2)), m.group(3)
        if not soff.startswith("s"):
            continue                      # immediate soffset: LLVM inserts the wait states itself
        total += 1
        nk, nxt = ins[i + 1]
        d = dst.match(nxt)
        if not d or d.group(1).startswith(("v_cmp", "v_mfma", "v_readfirstlane", "v_readlane")):
            continue
        a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
        if a <= hi and b >= lo:
            hits += 1
            print("%s: %s\n    %s\n    %s" % (path.split("/")[-1], k[:80], s, nxt))
print("%d wide buffer stores with an SGPR soffset, %d followed directly by a VALU write of their data registers" % (total, hits))
