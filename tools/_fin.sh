#!/bin/bash
O=gpurun_out/r6t; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 --save-profiles $O/profiles > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6t/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["kernel_ms"], d["config"]["build_id"])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d["kernel_ms_rocprof"].items() if k!="source"})
for k,v in d["secondary"].items():
    print(k, "ERROR "+v["error"] if "error" in v else (round(v.get("ms_per_step", v.get("ms_per_call", v.get("parcel_propagation_ms", v.get("LCS_call_ms", 0)))),3)))
PY
