"""Scan the compiler's output for the gfx950 hazard found in round 6 (csrc/sdf_value_split.hip): a v_mfma_f32_16x16x16_bf16 whose
accumulator input (srcC) is the destination of a v_mfma_f32_16x16x32_bf16 issued fewer than MIN_GAP instructions before it.  hipcc 7.2
counts the passes of the K = 32 shape as those of the K = 16 one, so the dependent MFMA of the SHORTER shape reads the accumulator before
the write has landed (same-shape back-to-back accumulation is forwarded by the hardware and is fine).
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only f.hip -o x.s;  python tools/scan_mfma_shape_hazard.py x.s ..."""
import re
import sys

MIN_GAP = 4          # independent instructions between the two; the builds that were wrong had 0-2
mfma = re.compile(r"^\s*(v_mfma_f32_16x16x(?:32|16)_bf16)\s+([va])\[(\d+):(\d+)\],\s*\S+?\],\s*\S+?\],\s*([va])\[(\d+):(\d+)\]")


def scan(text):
    ins = []
    kernel = "?"
    for ln in text.splitlines():
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        ins.append((kernel, s))
    pairs, hits = 0, []
    for i, (k, s) in enumerate(ins):
        m = mfma.match(s)
        if not m or "16x16x16" not in m.group(1):
            continue
        cfile, clo, chi = m.group(5), int(m.group(6)), int(m.group(7))
        for back in range(1, MIN_GAP + 1):
            if i - back < 0:
                break
            pk, ps = ins[i - back]
            pm = mfma.match(ps)
            if pm and "16x16x32" in pm.group(1) and pm.group(2) == cfile and int(pm.group(3)) <= chi and int(pm.group(4)) >= clo:
                hits.append((k, ps, s, back - 1))
                break
            if pm and pm.group(2) == cfile and int(pm.group(3)) <= chi and int(pm.group(4)) >= clo:
                break                                  # a same-shape writer of the accumulator lies in between: forwarded
        pairs += 1
    return pairs, hits


if __name__ == "__main__":
    total = bad = 0
    for path in sys.argv[1:]:
        n, hits = scan(open(path).read())
        total += n
        bad += len(hits)
        for k, a, b, gap in hits:
            print("%s: %s\n    %s\n    %s   (%d instructions between)" % (path.split("/")[-1], k[:70], a, b, gap))
    print("%d K=16 bf16 MFMAs, %d fed by a K=32 bf16 MFMA fewer than %d instructions earlier" % (total, bad, MIN_GAP))
