#!/bin/bash
# Register / LDS use of the advect kernels for a set of -D flags:  tools/regs.sh [-DFLAG ...]
mkdir -p build/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-pass-failed -S --cuda-device-only "$@" \
  -o build/isa/regs_tmp.s lagrangiancoherence_amd/csrc/advect.hip 2>/dev/null
awk '/^  - \.agpr_count/{a=1} /\.name:/{n=$2} /\.group_segment_fixed_size:/{l=$2} /\.sgpr_count:/{s=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "lds", l, "spill", $2}' build/isa/regs_tmp.s | grep -E "advect_lds_kernelILi[13]ELi4ELb1" 
