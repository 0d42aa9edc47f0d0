#!/bin/bash
# A round's closing record on the GPU box, on the build that ships: the GPU suite three ways (tools/gpu_check.sh), smoke(), the
# driver's own bench command with --save-profiles (its live rocprofv3 passes write profiles/rNN/c3_o1_*), the no-flag run.
# usage: tools/final_record.sh [outdir under gpurun_out]   then copy <outdir>/profiles/* and the two lines into profiles/rNN/
T=${1:-final}; O=gpurun_out/$T; mkdir -p $O
tools/gpu_check.sh $T || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --save-profiles $O/profiles > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-command rc=$?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - $O <<'PY'
import json, sys
for f in ("bench_driver","bench_default"):
    d=json.load(open(f"{sys.argv[1]}/{f}.json"))
    print(f, "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["kernel_ms"].items()}, d["config"]["build_id"])
    print("  ", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["kernel_ms_rocprof"].items() if k!="source"})
    for k,v in d["secondary"].items():
        print("   ", k, "ERROR "+v["error"] if "error" in v else round(v.get("ms_per_step", v.get("ms_per_call", v.get("parcel_propagation_ms", v.get("LCS_call_ms", 0)))),3), v.get("kernel",""))
PY
