#!/bin/bash
# round 6, call 4: interleaved partition (exchange form) on the GPU -- tests, shard costs -- and the host route's transfer options
set -o pipefail
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_sharded_gpu.py tests/test_bench_contract_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
timeout -k 10 300 python tools/shard_costs.py c4p c3sp > $O/shard_costs_interleaved.jsonl 2> $O/shard_costs.err; echo "shard rc=$?"
python - <<'PY'
import json
for ln in open("gpurun_out/r6d/shard_costs_interleaved.jsonl"):
    d=json.loads(ln); print(d["workload"], d["partition"], "pack", round(d["pack_ms"],3))
    for N,v in d["per_N"].items():
        print("  N", N, "step", v.get("step_ms_without_exchange"), "eff", v.get("efficiency"), "advect", [q["advect_ms"] for q in v["ranks"]], v.get("advect_spread"))
PY
timeout -k 10 120 ./build/host_route_probe > $O/host_route_probe.txt 2>&1; cat $O/host_route_probe.txt
timeout -k 10 200 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1; cat $O/pcie_rate.txt
