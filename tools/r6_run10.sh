#!/bin/bash
# round 6, call 10: GPU suite on the final kernels (deferred clamps in the one-seed order-1 kernel too), A/B of that kernel, default bench with --save-profiles
set -o pipefail
O=gpurun_out/r6j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
REPS=3 tools/ab.sh r6j/ab1 "--seeds 2048 --steps 10 --warmup 3 --no-secondary --no-live-counters --no-cpu-baseline" defer=lagrangiancoherence_amd/liblcs_hip.so r5form=build/libs/lds1_r5.so 2>&1 | grep json | tee $O/ab_one_seed.txt
timeout -k 10 600 python bench.py --save-profiles $O/profiles > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6j/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["kernel_ms"])
print("kernel_ms_rocprof", {k:v for k,v in d.get("kernel_ms_rocprof",{}).items() if k!="source"})
r=d["roofline"]; print("frac", r["frac"], "binding", r.get("binding",{}).get("frac"), "alg/hbm", r.get("algorithmic_over_hbm_peak"), "traffic", r.get("traffic"), "lim", r.get("limiting_unit"))
for k,v in d.get("secondary",{}).items(): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","ms_per_call","kernel","error","kernel_ms","LCS_call_ms","parcel_propagation_return_traj_ms","cpu_oracle_lcs_ms","traffic","hbm_traffic_frac","split_ms","serial_form_ms_per_call")})
print("cpu", d.get("cpu_baseline"))
PY
