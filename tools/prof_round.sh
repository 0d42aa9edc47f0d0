#!/bin/bash
# A round's profile collection for the non-default workloads: tools/prof_all.sh (every rocprofv3 guarded) per workload, on the GPU
# box; then tools/summarize_rNN.sh here.  usage: tools/prof_round.sh r06      Stops at the first workload whose passes fail.
P=${1:-r06}
log=gpurun_out/prof_$P.log
: > $log
run() { echo "$(date +%T) $*" >> $log; bash tools/prof_all.sh "$@" >> $log 2>&1 || { echo "FAILED $1" >> $log; tail -5 $log; exit 1; }; }
run ${P}_c3_o3 --order 3
run ${P}_c3_o1_traj --traj
run ${P}_c2 --workload c2
run ${P}_c2_o3 --workload c2 --order 3
run ${P}_c2_wind_f32 --workload c2 --wind-f32
echo "$(date +%T) done" >> $log
tail -5 $log
