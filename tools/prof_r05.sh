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
# (item 3's two levers -- LDS bank conflicts at pitch 24, fabric traffic with 2 tile rows per XCD chunk -- were collected once, on the
#  library of that experiment: profiles/r05/c2_p24_*, c2_x2_*; they concern the order-1 float64 advect kernel only)
echo "$(date +%T) done" >> $log
tail -5 $log
