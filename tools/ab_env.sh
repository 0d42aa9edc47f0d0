#!/bin/bash
# A/B timing of run-time knobs on the GPU box (same library, different environment).
# usage: tools/ab_env.sh <outdir> "<bench args>" name1="ENV1=a ENV2=b" name2="" ...   (each variant runs REPS times, alternating)
OUT=$1; shift
ARGS=$1; shift
REPS=${REPS:-2}
mkdir -p gpurun_out/$OUT
for rep in $(seq 1 $REPS); do
  for kv in "$@"; do
    name=${kv%%=*}; envs=${kv#*=}
    env $envs python bench.py --no-cpu-baseline $ARGS > gpurun_out/$OUT/${name}_$rep.json 2> gpurun_out/$OUT/${name}_$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/$OUT/${name}_$rep.err; }
  done
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/*.json"%sys.argv[1])):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "%.4g"%d["value"], d["roofline"]["kernel"], {k:round(v,3) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "unreadable", e)
PY
