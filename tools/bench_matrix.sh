#!/bin/bash
set -e
mkdir -p gpurun_out/mx
python bench.py > gpurun_out/mx/default.json 2> gpurun_out/mx/default.err
for k in 0 1 2; do python bench.py --no-cpu-baseline --settls $k > gpurun_out/mx/k$k.json 2>/dev/null; done
python bench.py --no-cpu-baseline --traj > gpurun_out/mx/traj.json 2>/dev/null
python bench.py --no-cpu-baseline --wind-scale 4 > gpurun_out/mx/w4.json 2>/dev/null
python bench.py --no-cpu-baseline --wind-scale 10 > gpurun_out/mx/w10.json 2>/dev/null
LCS_LDS_TILES=0 python bench.py --no-cpu-baseline > gpurun_out/mx/direct.json 2>/dev/null
LCS_LDS_TILES=0 python bench.py --no-cpu-baseline --wind-scale 10 > gpurun_out/mx/direct_w10.json 2>/dev/null
python bench.py --no-cpu-baseline --order 3 > gpurun_out/mx/o3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload c2 > gpurun_out/mx/c2.json 2>/dev/null
python tools/pcie_rate.py > gpurun_out/mx/pcie.txt 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/mx/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "%.4g"%d["value"], {k:round(v,3) for k,v in d["kernel_ms"].items()})
PY
tail -2 gpurun_out/mx/pcie.txt
