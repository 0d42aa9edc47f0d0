#!/bin/bash
O=gpurun_out/r6v; mkdir -p $O
nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for cfg in "T=-1 P=32" "T=3 P=32" "T=15 P=32" "T=31 P=32" "T=7 P=16" "T=7 P=8" "T=15 P=16" "T=15 P=8"; do
  eval $cfg
  if [ "$T" = "-1" ]; then unset LCS_HOST_THREADS; else export LCS_HOST_THREADS=$T; fi
  export LCS_HOST_PIECE_MB=$P
  echo "== threads $T piece $P MB"
  LCS_HOST_TIMING=1 python tools/pcie_rate.py > $O/t${T}_p${P}.log 2>&1
  grep "pass [234]" $O/t${T}_p${P}.log | awk '{print $4}' | tr '\n' ' '; echo
  grep -i "host_timing\|marks\|upload" $O/t${T}_p${P}.log | tail -2
done
