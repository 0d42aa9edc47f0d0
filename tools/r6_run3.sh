#!/bin/bash
# round 6, call 3: GPU suite (interleaved partition, prefilter restarts, guards), shard costs contiguous vs the product's interleaved
# chunks, default bench with the dispatch-trace based kernel_ms_rocprof
set -o pipefail
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
timeout -k 10 300 python tools/shard_costs.py c4 c4p c3s c3sp > $O/shard_costs.jsonl 2> $O/shard_costs.err; echo "shard rc=$?"
python - <<'PY'
import json
for ln in open("gpurun_out/r6c/shard_costs.jsonl"):
    d=json.loads(ln); print(d["workload"], "pack", round(d["pack_ms"],3))
    for N,v in d["per_N"].items():
        print("  N", N, "step", v.get("step_ms_without_exchange"), "eff", v.get("efficiency", v.get("efficiency_without_exchange")), "advect", [q["advect_ms"] for q in v["ranks"]], v.get("advect_spread"))
PY
timeout -k 10 400 python bench.py --save-profiles $O/profiles > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6c/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["kernel_ms"])
print("kernel_ms_rocprof", d.get("kernel_ms_rocprof"))
PY
