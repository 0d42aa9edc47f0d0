#!/bin/bash
# float64 order 1 on BASELINE configs[1]: LDS tile pitch (library variants) and tile rows per XCD chunk (LCS_XCD_CHUNK_ROWS)
out=${1:-gpurun_out/r5g}
mkdir -p $out
run() {  # name lib xcd_rows
  LCS_LIB=$2 LCS_XCD_CHUNK_ROWS=$3 python bench.py --workload c2 --steps 6 --warmup 2 --no-cpu-baseline > $out/$1.json 2> $out/$1.err || { echo "$1 FAILED"; tail -3 $out/$1.err; return; }
  python - $out/$1.json $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-16s"%sys.argv[2], 'ms_per_step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, d['roofline']['kernel'])
PY
}
P=$PWD/lagrangiancoherence_amd/liblcs_hip.so
for rep in 1 2; do
run p17_x1_$rep $P 1
run p18_x1_$rep $PWD/build/exp/lib_p18.so 1
run p20_x1_$rep $PWD/build/exp/lib_p20.so 1
run p24_x1_$rep $PWD/build/exp/lib_p24.so 1
run p17_x2_$rep $P 2
run p17_x4_$rep $P 4
run p17_x0_$rep $P 0
run p24_x2_$rep $PWD/build/exp/lib_p24.so 2
done
