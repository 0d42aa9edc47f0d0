"""Register budgets the kernels' performance depends on, read from the compiler's own metadata (hipcc -S, the product's
flags; no GPU).  The fused float64 prefilter (pack.hip: prefilter_fused_stream_kernel) holds a 40-row window of doubles per
lane and runs three waves per SIMD at 168 registers; a spilled window row turns a prefetched load into a load + wait + scratch
store at the head of every round (measured: 1.98 -> 3.5 ms with 12 registers spilled, 9 ms with 400) -- and the allocator
is touchy (loop strength reduction, the order of the LDS reads: see the kernel's comments).  So the budget is a test."""
import os
import re

import pytest

from tests.test_asm_hazards import CSRC, HIPCC, _assembly

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def _kernel_metadata(asm_path):
    """{kernel symbol: {field: int}} from the .amdgpu_metadata section of a hipcc -S listing."""
    text = open(asm_path).read()
    out = {}
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        fields = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|max_flat_workgroup_size):\s+(\d+)", block)}
        out[name.group(1)] = fields
    return out


def test_fused_prefilter_kernel_keeps_its_window_in_registers():
    md = _kernel_metadata(_assembly(os.path.join(CSRC, "pack.hip")))
    fused = {k: v for k, v in md.items() if "prefilter_fused_stream_kernel" in k}
    assert len(fused) == 2, sorted(md)                       # <double> and <float> (float32 wind, float64 coefficients)
    for name, f in fused.items():
        assert f["max_flat_workgroup_size"] == 768 and f["vgpr_count"] <= 168, (name, f)      # 12 waves: three per SIMD
        # at most ONE 4-byte loop-invariant spilled (reloaded once per round, away from the loads); no window row, no pointer
        assert f["vgpr_spill_count"] <= 1 and f["sgpr_spill_count"] == 0 and f["private_segment_fixed_size"] <= 8, (name, f)
        assert f["group_segment_fixed_size"] <= 16 * 1024, (name, f)


def test_streaming_sweeps_do_not_spill():
    md = _kernel_metadata(_assembly(os.path.join(CSRC, "pack.hip")))
    for name, f in md.items():
        if "prefilter_cols_stream_kernel" in name or "prefilter_rows_stream_kernel" in name or "pads_ext_kernel" in name or "pack_fused_kernel" in name:
            assert f["vgpr_spill_count"] == 0 and f["private_segment_fixed_size"] == 0, (name, f)
