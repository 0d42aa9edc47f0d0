#!/bin/bash
# bench.py --workload c2 at orders 1 and 3 (serial form) for library variants:  tools/ab_c2_libs.sh <outdir> name=lib.so ...
out=$1; shift
mkdir -p $out
for rep in 1 2; do for kv in "$@"; do name=${kv%%=*}; lib=${kv#*=}; for o in 1 3; do
  LCS_LIB=$PWD/$lib LCS_PIPELINE=0 python bench.py --workload c2 --order $o --steps 6 --warmup 2 --no-cpu-baseline > $out/${name}_o${o}_$rep.json 2> $out/${name}_o${o}_$rep.err || { echo "$name o$o FAILED"; tail -3 $out/${name}_o${o}_$rep.err; continue; }
  python - $out/${name}_o${o}_$rep.json ${name}_o${o}_$rep <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s"%sys.argv[2], 'ms_per_step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, d['roofline']['kernel'])
PY
done; done; done
