#!/bin/bash
# The GPU suite on the current build three ways (as at the end of round 5): plain, with every engine allocation NaN-poisoned
# (LCS_DEBUG_POISON=1: an element a kernel forgot to write cannot look right), and with the round-6 host-side defaults switched
# off (LCS_HOST_PIPELINE=0 LCS_HOST_CACHE=0: plain copies, allocate / free per call) next to the two-sweep prefilter fallback.
# usage (on the GPU box): tools/gpu_check.sh <outdir under gpurun_out>
O=gpurun_out/${1:-check}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_plain.log 2>&1; echo "plain rc=$?"; tail -2 $O/tests_plain.log
LCS_DEBUG_POISON=1 python -m pytest tests -m gpu -x -q > $O/tests_poison.log 2>&1; echo "poison rc=$?"; tail -2 $O/tests_poison.log
LCS_HOST_PIPELINE=0 LCS_HOST_CACHE=0 LCS_FUSED_PREFILTER=0 python -m pytest tests -m gpu -x -q > $O/tests_fallbacks.log 2>&1; echo "fallbacks rc=$?"; tail -2 $O/tests_fallbacks.log
