#!/bin/bash
# usage: tools/sigma_sweep.sh name=lib.so ...   (rocprofv3 kernel durations of the float32 sigma kernels, 4096^2 cells)
mkdir -p gpurun_out/sig
R=$PWD
for kv in "$@"; do
  name=${kv%%=*}; lib=${kv#*=}
  rm -rf $R/gpurun_out/sig/p_$name
  (cd /tmp && TMPDIR=/tmp LCS_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sig/p_$name -- python3 $R/tools/sigma_ab.py > $R/gpurun_out/sig/ab_$name.log 2>&1)
  echo "== $name: $(grep 'bitwise' gpurun_out/sig/ab_$name.log)"
  cat gpurun_out/sig/p_$name/*/*kernel_stats.csv | grep sigma | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   %-70s %8.1f us'%(r[0][:70], float(r[3])/1000))"
done
