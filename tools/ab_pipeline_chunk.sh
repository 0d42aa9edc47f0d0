#!/bin/bash
# float64 order 3 on BASELINE configs[1]: the pipelined form's chunk length (levels per pack / advect stage):  tools/ab_pipeline_chunk.sh <outdir> [chunk ...]
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for c in "$@"; do
    LCS_PIPELINE=1 LCS_PIPELINE_CHUNK=$c python bench.py --workload c2 --order 3 --steps 6 --warmup 2 --no-cpu-baseline > $out/chunk${c}_$rep.json 2> $out/chunk${c}_$rep.err || { echo "chunk $c FAILED"; tail -3 $out/chunk${c}_$rep.err; continue; }
    python - $out/chunk${c}_$rep.json $c <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("chunk", sys.argv[2], "step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms"].items()})
PY
  done
done
