#!/bin/bash
# round 6, call 8: C2 (float64, seeds = nodes) against the launch-shape knobs nobody swept on it; default bench with the host route case
set -o pipefail
O=gpurun_out/r6h; mkdir -p $O
REPS=2 tools/ab_env.sh r6h/c2 "--workload c2 --steps 10 --warmup 3" base="LCS_NONE=1" chunk16="LCS_LEVEL_CHUNK=16" chunk64="LCS_LEVEL_CHUNK=64" chunk100="LCS_LEVEL_CHUNK=100" chunk0="LCS_LEVEL_CHUNK=0" split8="LCS_XCD_SPLIT=8" order0="LCS_TILE_ORDER=0" order2="LCS_TILE_ORDER=2" 2>&1 | tee $O/c2_knobs.txt
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6h/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["kernel_ms"])
print(json.dumps(d["secondary"].get("c3 host route"), indent=0)[:1500])
PY
