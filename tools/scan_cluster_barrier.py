"""Scan clip_vit.hip's assembly: every arrival of the XCD-local cluster barrier (clip_cluster.hpp cluster_barrier: the global_atomic_add on
the cluster counter) must be preceded by a workgroup barrier that every wave enters with `s_waitcnt vmcnt(0)` -- walking BACK from that
s_barrier, the wait is met before any global / buffer store (the compiler itself only emits lgkmcnt(0) there, so without the
explicit wait the members' stores may not be in L2 when another CU sees the counter).
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only clip_vit.hip -o x.s;  python tools/scan_cluster_barrier.py x.s"""
import re
import sys


def scan(text):
    """(arrivals found, arrivals whose barrier is not guarded) over the cluster kernels of one assembly file"""
    lines = text.splitlines()
    kernel, found, bad = None, 0, []
    ins = []                                                   # (kernel, instruction) without labels / comments / directives
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        ins.append((kernel, s))
    for i, (k, s) in enumerate(ins):
        if not (k and "cluster_kernel" in k and s.startswith("global_atomic_add ") and "offset:128" in s):
            continue
        found += 1
        j = i - 1
        while j >= 0 and ins[j][0] == k and ins[j][1] != "s_barrier":
            j -= 1
        if j < 0 or ins[j][0] != k:
            bad.append((k, i, "no s_barrier in front"))
            continue
        ok = False
        for t in range(j - 1, max(j - 400, -1), -1):
            u = ins[t][1]
            if re.match(r"s_waitcnt\s+vmcnt\(0\)", u):
                ok = True
                break
            if re.match(r"(global|buffer|flat|scratch)_(store|atomic)", u) or u == "s_barrier":
                break
        if not ok:
            bad.append((k, i, "s_barrier reached without s_waitcnt vmcnt(0) after the last store"))
    return found, bad


if __name__ == "__main__":
    total = nbad = 0
    for path in sys.argv[1:]:
        f, b = scan(open(path).read())
        total += f
        nbad += len(b)
        for k, i, why in b:
            print("%s: %s: %s" % (path.split("/")[-1], k[:60], why))
    print("%d cluster-barrier arrivals, %d unguarded" % (total, nbad))
