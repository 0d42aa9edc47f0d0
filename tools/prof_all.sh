#!/bin/bash
# usage: prof_all.sh <tag> <bench args...>      kernel trace, traffic passes and the counter passes of one bench command, every
# rocprofv3 under tools/rocprof_guarded.sh; stops at the first pass that fails or is killed at its limit.
# (The default workload's summaries are written by the run itself: `python bench.py --save-profiles profiles/rNN`.)
set -e
R=$PWD
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
$R/tools/rocprof_guarded.sh --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-live-counters "$@" > $R/gpurun_out/$TAG/kt.log 2>&1
$R/tools/rocprof_guarded.sh --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-live-counters "$@" > $R/gpurun_out/$TAG/fetch.log 2>&1
$R/tools/rocprof_guarded.sh --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary --no-live-counters "$@" > $R/gpurun_out/$TAG/write.log 2>&1
cd $R
bash tools/pmc_pass.sh $TAG "$@"
python bench.py --no-secondary --no-live-counters "$@" > gpurun_out/$TAG/bench_stdout.json 2> gpurun_out/$TAG/bench_stderr.log
echo profiled $TAG
