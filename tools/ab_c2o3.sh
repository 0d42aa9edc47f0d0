#!/bin/bash
# float64 order 3 on BASELINE configs[1]: ext image (round 4) against no ext image (round 5), serial and pipelined
out=${1:-gpurun_out/r5b}
mkdir -p $out
for ext in 1 0; do for pipe in 0 1; do
  LCS_EXT_IMAGE=$ext LCS_PIPELINE=$pipe python bench.py --workload c2 --order 3 --steps 5 --warmup 2 --no-cpu-baseline > $out/c2o3_ext${ext}_pipe${pipe}.json 2> $out/c2o3_ext${ext}_pipe${pipe}.err || exit 1
  python - $out/c2o3_ext${ext}_pipe${pipe}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], 'ms_per_step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, d['roofline']['kernel'])
PY
done; done
