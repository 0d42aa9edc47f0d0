#!/bin/bash
# round 6, call 2: GPU suite on the build with the fixed-row prefilter restarts, the default bench run (new secondary cases, live
# kernel-trace pass, --save-profiles), configs[1] at order 3 (the pack whose restarts changed)
set -o pipefail
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -4 $O/tests.log
timeout -k 10 400 python bench.py --save-profiles $O/profiles > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6b/bench_default.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["kernel_ms"])
print("kernel_ms_rocprof", d.get("kernel_ms_rocprof"))
r=d["roofline"]; print("binding", r.get("binding"), "alg/hbm", r.get("algorithmic_over_hbm_peak"), "lim", r.get("limiting_unit"))
for k,v in d.get("secondary",{}).items(): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","kernel","error","kernel_ms","LCS_call_ms","parcel_propagation_return_traj_ms","cpu_oracle_lcs_ms","cpu_oracle_parcel_propagation_ms","traffic","hbm_traffic_frac","algorithmic_over_hbm_peak")})
PY
for i in 1 2; do timeout -k 10 200 python bench.py --workload c2 --order 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/c2o3_$i.json 2> $O/c2o3_$i.err; python -c "
import json;d=json.load(open('$O/c2o3_$i.json'));print('c2 o3', round(d['ms_per_step'],3), d['kernel_ms'])"; done
