#!/bin/bash
# Round 5's profile collection: tools/prof_all.sh for every workload in profiles/r05 (run on the GPU box; then tools/summarize_r05.sh here)
log=gpurun_out/prof_r05.log
: > $log
run() { echo "$(date +%T) $*" >> $log; bash tools/prof_all.sh "$@" >> $log 2>&1 || echo "FAILED $1" >> $log; }
run r05_c3_o1
run r05_c3_o3 --order 3
run r05_c3_o1_traj --traj
run r05_c2 --workload c2
run r05_c2_o3 --workload c2 --order 3
run r05_c2_wind_f32 --workload c2 --wind-f32
# item 3's two levers with the counters they were meant to move: LDS bank conflicts at pitch 24, fabric traffic with 2 tile rows per XCD chunk
echo "$(date +%T) pitch 24 counters" >> $log
LCS_LIB=$PWD/build/exp/lib_p24.so bash tools/pmc_pass.sh r05_c2_p24 --workload c2 >> $log 2>&1 || echo "FAILED p24" >> $log
echo "$(date +%T) xcd rows 2 traffic" >> $log
LCS_XCD_CHUNK_ROWS=2 bash tools/prof_all.sh r05_c2_x2 --workload c2 >> $log 2>&1 || echo "FAILED x2" >> $log
echo "$(date +%T) done" >> $log
tail -5 $log
