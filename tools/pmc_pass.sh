#!/bin/bash
# usage: pmc_pass.sh <tag> <bench args...> ; one guarded counter pass per line of tools/pmc_sets.txt, outputs under gpurun_out/<tag>/
# (a pass that fails or exceeds ROCPROF_LIMIT is reported and ENDS the script: no further GPU step after a killed one)
R=$PWD
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  $R/tools/rocprof_guarded.sh --pmc $line --output-format csv -d $R/gpurun_out/$TAG/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-live-counters "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1 \
    || { echo "pass $i failed (status $?): $line"; tail -5 $R/gpurun_out/$TAG/p$i.log; exit 1; }
done < $R/tools/pmc_sets.txt
echo done $i passes
