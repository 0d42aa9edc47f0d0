#!/bin/bash
# usage: prof_all.sh <tag> <bench args...>
set -e
R=$PWD
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/$TAG/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/$TAG/write.log 2>&1
cd $R
bash tools/pmc_pass.sh $TAG "$@"
python bench.py --no-secondary "$@" > gpurun_out/$TAG/bench_stdout.json 2> gpurun_out/$TAG/bench_stderr.log
echo profiled $TAG
