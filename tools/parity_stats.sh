#!/bin/bash
# The parity statistics of the full-size tests, committed:  tools/parity_stats.sh [profiles/r05/parity_stats.txt]
# Runs tests/test_gpu_fullsize*.py once with -s and keeps the lines that say how far the engine is from the float64
# answer next to how far the float32 ORACLE is ("gpu32 ... | oracle32 ..."), the float64 maxima, and the verified
# seam teleports -- the numbers the floors in those tests are judged against.
out=${1:-profiles/r05/parity_stats.txt}
mkdir -p $(dirname $out) gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c345.py -s -q -m gpu > gpurun_out/parity_stats_raw.log 2>&1
rc=$?
{
  echo "# tools/parity_stats.sh: pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c345.py -s -m gpu  (rc $rc)"
  echo "# library build $(python -c 'from lagrangiancoherence_amd import _capi; print(_capi.load().lc_build_id().decode())' 2>/dev/null)"
  echo "# errors in degrees (positions) or relative (sigma) against the float64 oracle; 'oracle32' = the float32 oracle's own error"
  grep -E "gpu32 median|max \|dx\||teleport|passed|failed" gpurun_out/parity_stats_raw.log
} > $out
cat $out
exit $rc
