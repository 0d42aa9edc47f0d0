#!/bin/bash
# round 6, call 6: direct levels in the two-seed kernel (tests, A/B, counters), the pipelined host route (tests, rate)
set -o pipefail
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
REPS=3 tools/ab.sh r6f/ab "--steps 10 --warmup 3 --no-secondary --no-live-counters" dl7=lagrangiancoherence_amd/liblcs_hip.so dl0=build/libs/dl0.so dl3=build/libs/dl3.so dl15=build/libs/dl15.so 2>&1 | tee $O/ab.txt
for i in 1 2 3; do timeout -k 10 120 python tools/pcie_rate.py 2>/dev/null | tail -2; done | tee $O/pcie_rate.txt
LCS_HOST_PIPELINE=0 timeout -k 10 120 python tools/pcie_rate.py 2>/dev/null | tail -2 | tee $O/pcie_rate_serial.txt
timeout -k 10 300 python tools/shard_costs.py c4 c4p > $O/shard_costs.jsonl 2> $O/shard_costs.err
python - <<'PY'
import json
for ln in open("gpurun_out/r6f/shard_costs.jsonl"):
    d=json.loads(ln); print(d["workload"], d.get("partition"), "pack", round(d["pack_ms"],3))
    for N,v in d["per_N"].items():
        print("  N", N, "step", v.get("step_ms_without_exchange"), "eff", v.get("efficiency", v.get("efficiency_without_exchange")), "advect", [q["advect_ms"] for q in v["ranks"]], v.get("advect_spread"))
PY
