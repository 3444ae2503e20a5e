"""gfx950 store-data hazard that the compiler does not cover (profiles/r05_store_hazard.txt, tools/micro/store_war_hazard.hip): no kernel
of the library may contain a buffer_store_dwordx3 / x4 with an SGPR soffset that is followed DIRECTLY by a VALU write of its data
registers.  Compiles the files that issue such stores to assembly (hipcc cross-compiles without a GPU) and scans them."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "shapeclipper_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_wide_buffer_store_is_followed_directly_by_a_write_of_its_data(tmp_path):
    # only files that can emit raw buffer stores: the MLP chain kernels (tbl_store) and whoever else names the builtin
    files = [f for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")))
             if re.search(r"raw_buffer_store|mlp_tile\.hpp|mlp_xch\.hpp|rgb_common\.hpp|mlp_presplit\.hpp", open(f).read())]
    assert len(files) >= 7 and any(f.endswith("sdf_value_split.hip") for f in files), files

    def build(f):
        out = str(tmp_path / (os.path.basename(f)[:-4] + ".s"))
        extra = ["-ffp-contract=off"] if os.path.basename(f) == "render.hip" else []
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), *extra, "-S",
                            "--cuda-device-only", f, "-o", out], capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, r.stderr[-2000:]
        return out
    with ThreadPoolExecutor(4) as ex:
        asm = list(ex.map(build, files))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py"), *asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"(\d+) wide buffer stores with an SGPR soffset, (\d+) followed directly", last)
    assert m, r.stdout[-1000:]
    assert int(m.group(1)) >= 100, last           # the scan saw the stores it is about
    assert int(m.group(2)) == 0, r.stdout[-3000:]


def test_the_scanner_sees_the_pattern(tmp_path):
    """The two lines hipcc produced for sdf_fwd's training instance are reported; the same store with an instruction in between, with an
    immediate soffset, or followed by a read of its data is not."""
    asm = tmp_path / "x.s"
    asm.write_text("\n".join([
        "_Z1kv:",
        "\tbuffer_store_dwordx4 v[90:93], v141, s[8:11], s74 offen",
        "\tv_mul_f32_e32 v90, v91, v1",                                  # hazard
        "\tbuffer_store_dwordx4 v[82:85], v141, s[8:11], s75 offen",
        "\ts_nop 0",
        "\tv_mul_f32_e32 v82, v83, v1",                                  # one instruction in between: fine
        "\tbuffer_store_dwordx4 v[40:43], v141, s[8:11], 0 offen offset:1024",
        "\tv_mov_b32_e32 v40, v1",                                       # immediate soffset: the compiler's own wait states cover it
        "\tbuffer_store_dwordx4 v[20:23], v141, s[8:11], s73 offen",
        "\tv_mul_f32_e32 v0, v20, v1",                                   # reads the data, writes elsewhere
        "\tbuffer_store_dwordx3 v[10:12], v141, s[8:11], s73 offen",
        "\tv_pk_mul_f32 v[12:13], v[2:3], v[4:5]",                       # hazard (overlaps v12)
        "\ts_endpgm"]) + "\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_store_hazard.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("4 wide buffer stores with an SGPR soffset, 2 followed directly"), r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_cluster_barrier_arrivals_wait_for_their_stores(tmp_path):
    """the XCD-local cluster barrier of the CLIP tower publishes with a relaxed counter increment; every wave must therefore
    enter the workgroup barrier in front of it with `s_waitcnt vmcnt(0)` (its att / h / x stores acknowledged by L2).  The compiler does not
    emit that wait by itself; the explicit one in clip_cluster.hpp must survive in the ISA of both instances of the kernel."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_cluster_barrier
    out = str(tmp_path / "clip_vit.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(CSRC, "clip_vit.hip"), "-o", out], capture_output=True, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stderr[-2000:]
    found, bad = scan_cluster_barrier.scan(open(out).read())
    assert found >= 8 and not bad, bad
    # the scanner reports the unguarded form
    f2, b2 = scan_cluster_barrier.scan("\n".join([
        "_ZN2sc2cl26clip_layers_cluster_kernelILb0EEEvv:", "\tglobal_store_dwordx4 v1, v[2:5], s[0:1]", "\ts_waitcnt lgkmcnt(0)", "\ts_barrier",
        "\tglobal_atomic_add v34, v1, s[94:95] offset:128", "\ts_endpgm"]))
    assert f2 == 1 and len(b2) == 1


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_short_mfma_is_fed_by_a_k32_mfma_directly_in_front_of_it(tmp_path):
    """Round 6 (csrc/sdf_value_split.hip): a v_mfma_f32_16x16x16_bf16 issued directly behind the v_mfma_f32_16x16x32_bf16 that writes its
    accumulator read the accumulator too early (level grid wrong by 0.4).  No kernel that uses both shapes may contain the pair fewer
    than 4 instructions apart (tools/scan_mfma_shape_hazard.py); the scanner reports the failing order."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_mfma_shape_hazard as S
    seen = 0
    for name in ("sdf_value_split.hip", "rgb_fwd.hip", "sdf_fwd_stream.hip", "rgb_bwd.hip"):          # every file that issues both shapes (mlp_presplit.hpp)
        out = str(tmp_path / (name[:-4] + ".s"))
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                            os.path.join(CSRC, name), "-o", out], capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, r.stderr[-2000:]
        n, hits = S.scan(open(out).read())
        assert n >= 20 and not hits, (name, hits[:3])
        seen += n
    assert seen >= 300
    assert sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and "mlp_presplit.hpp" in open(os.path.join(CSRC, f)).read()) == \
        ["rgb_bwd.hip", "rgb_fwd.hip", "sdf_fwd_stream.hip", "sdf_value_split.hip"]                    # a new user of the header must be added to the list above
    n2, hits2 = S.scan("\n".join([
        "_Z1kv:",
        "\tv_mfma_f32_16x16x32_bf16 v[28:31], v[24:27], v[0:3], v[28:31]",
        "\tv_sub_f32_e32 v26, v16, v12",
        "\tv_mfma_f32_16x16x16_bf16 v[28:31], v[56:57], v[94:95], v[28:31]",                # hazard: 1 instruction between
        "\tv_mfma_f32_16x16x32_bf16 v[40:43], v[24:27], v[0:3], v[40:43]",
        "\tv_mfma_f32_16x16x16_bf16 v[44:47], v[56:57], v[94:95], v[44:47]",                # another accumulator: fine
        "\tv_mfma_f32_16x16x16_bf16 v[44:47], v[58:59], v[94:95], v[44:47]",                # same shape back to back: forwarded
        "\ts_endpgm"]))
    assert n2 == 3 and len(hits2) == 1
