#!/bin/bash
# round 6, call 1: GPU suite on the new default, then A/B of the two-seed order-1 kernel's variants, then the redo statistics
set -o pipefail
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/tests.log 2>&1; echo "tests rc=$?" | tee gpurun_out/r6a/tests.rc
tail -5 gpurun_out/r6a/tests.log
REPS=3 tools/ab.sh r6a/ab "--steps 10 --warmup 3 --no-secondary --no-live-counters" new=lagrangiancoherence_amd/liblcs_hip.so r5form=build/libs/r5form.so t12x10=build/libs/t12x10.so t14x9=build/libs/t14x9.so 2>&1 | tee gpurun_out/r6a/ab.txt
LCS_LIB=$PWD/build/libs/stamps.so timeout -k 10 300 python tools/dbg_stamps.py > gpurun_out/r6a/stamps_16x8.txt 2>&1
LCS_LIB=$PWD/build/libs/t12x10s.so timeout -k 10 300 python tools/dbg_stamps.py > gpurun_out/r6a/stamps_12x10.txt 2>&1
cat gpurun_out/r6a/stamps_16x8.txt gpurun_out/r6a/stamps_12x10.txt
