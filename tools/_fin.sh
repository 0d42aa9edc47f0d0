#!/bin/bash
O=gpurun_out/r6fin; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bench_contract_gpu.py -x -q > $O/contract.log 2>&1; echo "contract rc=$?"; tail -n 2 $O/contract.log
python bench.py --gpus 1 --steps 20 --warmup 5 --save-profiles $O/profiles > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver-command rc=$?"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
for f in ("bench_driver","bench_default"):
    d=json.load(open(f"gpurun_out/r6fin/{f}.json"))
    print(f, "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["kernel_ms"].items()}, d["config"]["build_id"])
    print("  ", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["kernel_ms_rocprof"].items() if k!="source"})
    print("  ", d["roofline"].get("binding"), d["roofline"].get("limiting_unit",{}).get("valu_instr_per_wave_timestep"))
    for k,v in d["secondary"].items():
        print("   ", k, "ERROR "+v["error"] if "error" in v else round(v.get("ms_per_step", v.get("ms_per_call", 0)),3), v.get("kernel",""))
PY
