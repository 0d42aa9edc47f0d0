"""The inline-asm blocks of the HIP kernels against the gfx950 hazards hipcc cannot see inside them.

hipcc inserts the wait states gfx950 needs between ordinary instructions, but not when the CONSUMER is a VALU
instruction hidden in an asm block (tools/asm_hazard_probe.hip shows the two rules it misses:
VALU-written SGPR -> VALU read, 2 wait states; trans result -> VALU read, 1).  A violated rule reads a stale register --
a wrong LDS address, a wrong tile anchor -- intermittently, depending on what else the SIMD issued in between.
tools/asm_hazards.py walks the compiler's assembly of every kernel and follows each asm instruction's sources
backwards through the hazard window (across basic-block entries too).  Runs on the CPU: hipcc -S cross-compiles."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "lagrangiancoherence_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-function", "-Wno-pass-failed"]


def _assembly(src: str) -> str:
    """hipcc -S of one translation unit with the product's flags, cached under build/asm by the sources' content."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)) + [os.path.join(ROOT, "include", "lcs_hip.h")]:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    out = os.path.join(ROOT, "build", "asm", f"{os.path.basename(src)}.{h.hexdigest()[:12]}.s")
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", out + ".tmp", src], check=True, capture_output=True)
        os.replace(out + ".tmp", out)
        # one listing per translation unit: the listings of earlier source states go (advect.hip's is 28 MB, and the tree
        # travels to the GPU box as a snapshot with a size limit)
        import glob
        for old in glob.glob(os.path.join(os.path.dirname(out), os.path.basename(src) + ".*.s")):
            if old != out:
                os.remove(old)
    return out


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def test_the_audit_finds_the_hazards_hipcc_leaves_in_the_probe(tmp_path):
    import asm_hazards
    out = str(tmp_path / "probe.s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "tools", "asm_hazard_probe.hip")], check=True, capture_output=True)
    sites, found = asm_hazards.audit(out)
    kinds = sorted((f["kernel"], f["kind"].split(" (")[0]) for f in found)
    # if a newer compiler closes one of these, the rule can be dropped from the audit -- not silently
    assert kinds == [("_Z20asm_valu_after_transPKfPi", "trans result read by asm VALU"),
                     ("_Z25asm_valu_after_sgpr_writePKfPii", "VALU-written SGPR read by asm VALU")], found
    assert sites == 3


def test_the_audit_knows_the_dpp_rules(tmp_path):
    """Round 6 put two v_mov_b32_dpp into the two-seed kernel's staging: a DPP instruction needs two wait states after a VALU
    write of its source and five after a VALU write of EXEC.  Hand-written listings: the audit flags the bare forms and accepts
    the form the kernel uses (the asm statement carries `s_nop 4`)."""
    import asm_hazards

    def listing(pre, nop):
        body = ["_Z4demov:", *("\t" + x for x in pre), "\t;;#ASMSTART", *(["\ts_nop %d" % nop] if nop is not None else []),
                "\tv_mov_b32_dpp v3, v10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", "\t;;#ASMEND", "\ts_endpgm"]
        f = tmp_path / "demo.s"
        f.write_text("\n".join(body) + "\n")
        return asm_hazards.audit(str(f))[1]
    kinds = lambda found: sorted(x["kind"].split(" (")[0] for x in found)
    assert kinds(listing(["v_add_f32_e32 v10, v1, v2"], None)) == ["VALU-written VGPR read by asm DPP"]
    assert kinds(listing(["v_add_f32_e32 v10, v1, v2", "s_mov_b32 s4, 0"], None)) == ["VALU-written VGPR read by asm DPP"]
    assert kinds(listing(["v_add_f32_e32 v10, v1, v2"], 1)) == []
    assert kinds(listing(["v_cmpx_gt_f32_e32 v1, v2", "s_mov_b32 s4, 0", "s_mov_b32 s5, 0"], None)) == ["VALU-written EXEC before asm DPP"]
    assert kinds(listing(["v_cmpx_gt_f32_e32 v1, v2"], 1)) == ["VALU-written EXEC before asm DPP"]
    assert kinds(listing(["v_cmpx_gt_f32_e32 v1, v2", "v_add_f32_e32 v10, v1, v2"], 4)) == []


def test_only_advect_hip_holds_inline_asm_instructions():
    """The audit below covers advect.hip; no other translation unit may grow an asm instruction unaudited."""
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")) and f != "advect.hip":
            src = open(os.path.join(CSRC, f)).read()
            assert not re.search(r'\basm\s*(volatile)?\s*\(\s*"[^"]*[a-z]', src), f


def test_no_hidden_hazard_at_any_asm_site_of_the_advect_kernels():
    import asm_hazards
    sites, found = asm_hazards.audit(_assembly(os.path.join(CSRC, "advect.hip")))
    assert sites > 1000            # v_cvt_flr_i32_f32 / v_mad_u32_u24 in every float32 kernel instance
    assert found == [], "\n".join(map(str, found[:20]))
