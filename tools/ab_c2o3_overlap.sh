#!/bin/bash
# Does the pack's LDS-hungry row sweep starve behind the order-3 float64 advect kernel (4 workgroups x 39 KB = all of a CU's LDS)?
# bench.py --workload c2 --order 3, pipelined (product) and serial, for library variants: tools/ab_c2o3_overlap.sh <outdir> name=lib ...
out=$1; shift
mkdir -p $out
for rep in 1 2; do for kv in "$@"; do name=${kv%%=*}; lib=${kv#*=}; for pipe in 1 0; do
  LCS_LIB=$PWD/$lib LCS_PIPELINE=$pipe python bench.py --workload c2 --order 3 --steps 6 --warmup 2 --no-cpu-baseline > $out/${name}_p${pipe}_$rep.json 2> $out/${name}_p${pipe}_$rep.err || { echo "$name FAILED"; tail -3 $out/${name}_p${pipe}_$rep.err; continue; }
  python - $out/${name}_p${pipe}_$rep.json ${name}_pipe${pipe}_$rep <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-26s"%sys.argv[2], 'ms_per_step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()})
PY
done; done; done
