#!/bin/bash
# float64 order 3 on BASELINE configs[1]: library variants (build/exp/lib_<name>.so), serial form, ext image on unless noted
out=${1:-gpurun_out/r5c}; shift
mkdir -p $out
run() {  # name lib ext
  LCS_LIB=$2 LCS_EXT_IMAGE=$3 LCS_PIPELINE=0 python bench.py --workload c2 --order 3 --steps 5 --warmup 2 --no-cpu-baseline > $out/$1.json 2> $out/$1.err || { echo "$1 FAILED"; tail -3 $out/$1.err; return; }
  python - $out/$1.json $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'ms_per_step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, d['roofline']['kernel'])
PY
}
P=$PWD/lagrangiancoherence_amd/liblcs_hip.so
run product_ext1 $P 1
run product_ext0 $P 0
for v in "$@"; do
  case $v in cubpf) run $v $PWD/build/exp/lib_$v.so 0;; *) run $v $PWD/build/exp/lib_$v.so 1;; esac
done
run product_ext1_again $P 1
