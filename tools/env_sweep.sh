#!/bin/bash
# usage: tools/env_sweep.sh <outdir> <lib.so> "<bench args>" VAR "v1 v2 ..." ; one bench run per value of the env variable VAR (REPS repeats, default 2)
OUT=$1; LIB=$2; ARGS=$3; VAR=$4; VALS=$5
REPS=${REPS:-2}
mkdir -p gpurun_out/$OUT
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env LCS_LIB=$PWD/$LIB $VAR=$v python bench.py --no-cpu-baseline --no-secondary $ARGS > gpurun_out/$OUT/${VAR}_${v}_$rep.json 2> gpurun_out/$OUT/${VAR}_${v}_$rep.err || { echo "$v FAILED"; tail -n 3 gpurun_out/$OUT/${VAR}_${v}_$rep.err; }
  done
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/*.json"%sys.argv[1])):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "%.4g"%d["value"], d["roofline"]["kernel"], {k:round(v,3) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "unreadable", e)
PY
