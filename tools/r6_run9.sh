#!/bin/bash
# round 6, call 9: level-chunk sweeps (C2 both orders, C3), snake tile order on C4's rank blocks, the host route with the buffer cache
set -o pipefail
O=gpurun_out/r6i; mkdir -p $O
REPS=2 tools/ab_env.sh r6i/c2 "--workload c2 --steps 10 --warmup 3 --no-cpu-baseline" c32="LCS_NONE=1" c8="LCS_LEVEL_CHUNK=8" c12="LCS_LEVEL_CHUNK=12" c16="LCS_LEVEL_CHUNK=16" c20="LCS_LEVEL_CHUNK=20" c24="LCS_LEVEL_CHUNK=24" c16o0="LCS_LEVEL_CHUNK=16 LCS_TILE_ORDER=0" 2>&1 | grep json | tee $O/c2_chunks.txt
REPS=2 tools/ab_env.sh r6i/c2o3 "--workload c2 --order 3 --steps 10 --warmup 3 --no-cpu-baseline" c32="LCS_NONE=1" c12="LCS_LEVEL_CHUNK=12" c16="LCS_LEVEL_CHUNK=16" c24="LCS_LEVEL_CHUNK=24" 2>&1 | grep json | tee $O/c2o3_chunks.txt
REPS=2 tools/ab_env.sh r6i/c3 "--steps 10 --warmup 3 --no-secondary --no-live-counters --no-cpu-baseline" c32="LCS_NONE=1" c16="LCS_LEVEL_CHUNK=16" c24="LCS_LEVEL_CHUNK=24" c48="LCS_LEVEL_CHUNK=48" 2>&1 | grep json | tee $O/c3_chunks.txt
LCS_TILE_ORDER=3 timeout -k 10 200 python tools/shard_costs.py c4 > $O/shard_costs_snake.jsonl 2> $O/shard.err
timeout -k 10 200 python tools/shard_costs.py c4 > $O/shard_costs_default.jsonl 2>> $O/shard.err
python - <<'PY'
import json
for f in ("snake","default"):
    for ln in open("gpurun_out/r6i/shard_costs_%s.jsonl"%f):
        d=json.loads(ln); print(f, d["workload"], "pack", round(d["pack_ms"],3))
        for N,v in d["per_N"].items():
            print("  N", N, "step", v.get("step_ms_without_exchange"), "eff", v.get("efficiency_without_exchange"), "advect", [q["advect_ms"] for q in v["ranks"]])
PY
LCS_HOST_TIMING=1 timeout -k 10 120 python tools/pcie_rate.py 2>&1 | grep "lc_lcs_host" | tail -8 | tee $O/pcie.txt
