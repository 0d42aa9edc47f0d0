#!/bin/bash
# round 6, call 5: XCD split for the interleaved chunks, redo histogram, bench contract tests
set -o pipefail
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests/test_sharded_gpu.py tests/test_bench_contract_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
for sp in 8; do
LCS_XCD_SPLIT=$sp timeout -k 10 300 python tools/shard_costs.py c4p c4 > $O/shard_costs_split$sp.jsonl 2> $O/shard_costs.err; echo "shard rc=$?"
python - $sp <<'PY'
import json,sys
for ln in open("gpurun_out/r6e/shard_costs_split%s.jsonl"%sys.argv[1]):
    d=json.loads(ln); print("split",sys.argv[1],d["workload"], d.get("partition"), "pack", round(d["pack_ms"],3))
    for N,v in d["per_N"].items():
        print("  N", N, "step", v.get("step_ms_without_exchange"), "eff", v.get("efficiency", v.get("efficiency_without_exchange")), "advect", [q["advect_ms"] for q in v["ranks"]], v.get("advect_spread"))
PY
done
LCS_LIB=$PWD/build/libs/stamps.so timeout -k 10 300 python tools/dbg_stamps.py > $O/stamps.txt 2>&1; tail -12 $O/stamps.txt
