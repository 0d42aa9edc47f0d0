#!/bin/bash
# usage: pmc_env.sh <tag> VAR=value <bench args...> ; HBM traffic and L2 hit/miss counters of one environment setting
R=$PWD
TAG=$1; KV=$2; shift; shift
mkdir -p $R/gpurun_out/$TAG
export "$KV"
cd /tmp && export TMPDIR=/tmp
i=0
for line in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/$TAG/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1 || { echo "pass $i failed: $line"; tail -n 5 $R/gpurun_out/$TAG/p$i.log; }
done
cd $R && echo "$KV" && python3 tools/pmc_sum.py $TAG | tail -n 30
