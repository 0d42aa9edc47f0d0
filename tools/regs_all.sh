#!/bin/bash
# Register / LDS / spill table of EVERY advect kernel (demangled), for a set of -D flags:  tools/regs_all.sh out.txt [-DFLAG ...]
out=$1; shift
mkdir -p build/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-pass-failed -S --cuda-device-only "$@" \
  -o build/isa/regs_all.s lagrangiancoherence_amd/csrc/advect.hip 2>/dev/null
awk '/\.name:/{n=$2} /\.group_segment_fixed_size:/{l=$2} /\.sgpr_count:/{s=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "lds", l, "spill", $2}' build/isa/regs_all.s \
  | while read n rest; do echo "$(echo $n | c++filt) $rest"; done | sort > "$out"
wc -l "$out"
