#!/bin/bash
# usage: pmc_pass.sh <tag> <bench args...> ; runs several counter passes, outputs under gpurun_out/<tag>/
set -e
R=$PWD
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/$TAG/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1 || { echo "pass $i failed: $line"; tail -5 $R/gpurun_out/$TAG/p$i.log; }
done < $R/tools/pmc_sets.txt
echo done $i passes
