#!/bin/bash
# A/B timing of library variants on the GPU box.
# usage: tools/ab.sh <outdir> "<bench args>" name1=path1.so name2=path2.so ...
set -e
OUT=$1; shift
ARGS=$1; shift
mkdir -p gpurun_out/$OUT
for kv in "$@"; do
  name=${kv%%=*}; lib=${kv#*=}
  LCS_LIB=$PWD/$lib python bench.py --no-cpu-baseline $ARGS > gpurun_out/$OUT/$name.json 2> gpurun_out/$OUT/$name.err || { echo "$name FAILED"; tail -3 gpurun_out/$OUT/$name.err; }
done
python - "$OUT" <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/*.json"%sys.argv[1])):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "%.4g"%d["value"], {k:round(v,3) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "unreadable", e)
PY
