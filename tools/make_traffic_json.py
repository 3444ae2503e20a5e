"""profiles/r06_traffic.json from the two PMC summaries (tools/summarize_pmc.py output of the FETCH_SIZE and WRITE_SIZE passes of
tools/perf_render.py --B 32 --iters 1): per entry point of the main render the HBM-side bytes of its largest dispatch, corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE counts the 128-byte requests of wide coalesced reads as 64 bytes -> doubled; WRITE_SIZE as is).
    python tools/make_traffic_json.py pmc_fetch_size.txt pmc_write_size.txt > profiles/r06_traffic.json"""
import json
import re
import sys

KERNEL_TO_ENTRY = [            # kernel name prefix -> bench.py's stable entry-point name
    ("sc::sdf_bwdw_kernel", "sc_sdf_backward_fused"),
    ("sc::st::sdf_fwd_stream_kernel", "sc_sdf_forward"), ("sc::sdf_fwd_kernel", "sc_sdf_forward"),
    ("sc::rgb_composite_fwd_split_kernel", "sc_rgb_composite_forward"), ("sc::rgb_composite_fwd_kernel", "sc_rgb_composite_forward"),
    ("sc::rgb_composite_bwd_kernel", "sc_rgb_composite_backward"),
]


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+%s=([0-9.e+-]+) GB" % counter, line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
kernels = {}
for prefix, entry in KERNEL_TO_ENTRY:
    names = [n for n in fetch if n.startswith(prefix)]
    if not names or entry in kernels:
        continue
    n = max(names, key=lambda k: fetch[k])
    f, w = fetch[n], write.get(n, 0.0)
    kernels[entry] = dict(kernel=n, fetch_size_gb=round(f, 4), write_size_gb=round(w, 4), traffic_bytes=int(round((2 * f + w) * 1e9, -5)))
print(json.dumps({
    "_comment": "HBM-side bytes per launch of the main render (B=32, 1,048,576 points) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                "tools/prof_r06.sh -> tools/perf_render.py --B 32 --iters 1; largest dispatch of each kernel; profiles/r06_pmc_fetch_size.txt, "
                "r06_pmc_write_size.txt, collected at the end of round 6 with the round's default kernels: streamed pre-split SDF forward, pre-split "
                "RGB forward). Correction per MI355X_MICROARCH.md: FETCH_SIZE counts the 128-B requests of wide coalesced reads as 64 B -> doubled; "
                "WRITE_SIZE uncorrected. Infinity-Cache hits are counted, so this is an upper bound on DRAM traffic.",
    "kernels": kernels,
    "render_total_bytes": sum(k["traffic_bytes"] for k in kernels.values())}, indent=1))
