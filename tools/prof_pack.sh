#!/bin/bash
# Kernel trace + counter passes (only sets that tools/pmc_sets.txt has run with: an unknown counter name aborts rocprofv3 and hangs) of lc_field_pack alone:  tools/prof_pack.sh <outdir under gpurun_out> [c2|c3] [order]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; W=${2:-c2}; O=${3:-3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/pack_bench.py $W $O 5 > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f"
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 $R/tools/pack_bench.py $W $O 2 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr
FETCH_SIZE
WRITE_SIZE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SETS
python3 $R/tools/pmc_by_kernel.py $OUT prefilter pads_ext 2>/dev/null | head -120
