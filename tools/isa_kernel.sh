#!/bin/bash
# Common-path instruction histogram of one advect kernel of the working tree (hipcc -S, ~75 s).
# usage: tools/isa_kernel.sh <mangled-name fragment> [extra -D flags ...]      e.g.  tools/isa_kernel.sh advect_lds2_kernelILi4ELb1ELi0E
set -e
K=$1; shift
mkdir -p build/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -fno-gpu-rdc -Iinclude "$@" \
    -o build/asm/advect_wt.s lagrangiancoherence_amd/csrc/advect.hip > /dev/null 2>&1
python tools/isa_hist.py build/asm/advect_wt.s "$K"
awk -v k="$K" '$0 ~ "^_Z" && index($0, k) {f=1} f{print} /^.Lfunc_end/{if(f)exit}' build/asm/advect_wt.s > build/asm/kernel_wt.s
grep -m4 "NumVgprs\|NumSgprs\|Occupancy\|ScratchSize" <(awk -v k="$K" 'index($0, k) && /^_Z/ {f=1} f' build/asm/advect_wt.s)
