#!/bin/bash
# Reduce gpurun_out/r05_* (tools/prof_all.sh) to the summaries under profiles/r05/, stamped with the build id of the library that ran.
set -e
mkdir -p profiles/r05
for o in 1 3; do
  W="{\"workload\":\"c3\",\"seeds\":4096,\"nt\":97,\"order\":$o,\"K\":4,\"dtype\":\"f32\"}"
  python profiles/summarize.py gpurun_out/r05_c3_o$o profiles/r05/c3_o$o "$W" > /dev/null
  python profiles/summarize_sq.py gpurun_out/r05_c3_o$o profiles/r05/c3_o$o "$W" 32 > /dev/null
  cp gpurun_out/r05_c3_o$o/bench_stdout.json profiles/r05/c3_o${o}_bench_stdout.json
done
W='{"workload":"c3","seeds":4096,"nt":97,"order":1,"K":4,"dtype":"f32","variant":true}'
python profiles/summarize.py gpurun_out/r05_c3_o1_traj profiles/r05/c3_o1_traj "$W" > /dev/null
python profiles/summarize_sq.py gpurun_out/r05_c3_o1_traj profiles/r05/c3_o1_traj "$W" 32 > /dev/null
cp gpurun_out/r05_c3_o1_traj/bench_stdout.json profiles/r05/c3_o1_traj_bench_stdout.json
W='{"workload":"c2","order":1,"K":4,"dtype":"f64","fuse_levels":true}'
python profiles/summarize.py gpurun_out/r05_c2 profiles/r05/c2 "$W" > /dev/null
python profiles/summarize_sq.py gpurun_out/r05_c2 profiles/r05/c2 "$W" 28.5714   # 200 levels in 7 launches (6 x 32 + 8) > /dev/null
cp gpurun_out/r05_c2/bench_stdout.json profiles/r05/c2_bench_stdout.json
W='{"workload":"c2","order":3,"K":4,"dtype":"f64","fuse_levels":true}'
python profiles/summarize.py gpurun_out/r05_c2_o3 profiles/r05/c2_o3 "$W" > /dev/null
python profiles/summarize_sq.py gpurun_out/r05_c2_o3 profiles/r05/c2_o3 "$W" 28.5714 > /dev/null
cp gpurun_out/r05_c2_o3/bench_stdout.json profiles/r05/c2_o3_bench_stdout.json
W='{"workload":"c2","order":1,"K":4,"dtype":"f64","fuse_levels":true,"wind":"f32"}'
python profiles/summarize.py gpurun_out/r05_c2_wind_f32 profiles/r05/c2_wind_f32 "$W" > /dev/null
python profiles/summarize_sq.py gpurun_out/r05_c2_wind_f32 profiles/r05/c2_wind_f32 "$W" 28.5714 > /dev/null
cp gpurun_out/r05_c2_wind_f32/bench_stdout.json profiles/r05/c2_wind_f32_bench_stdout.json
tools/regs.sh > /dev/null
python tools/isa_hist.py build/isa/regs_tmp.s advect_lds2_kernelILi4ELb1ELi0E --json profiles/r05/isa_hist_advect_lds2_k4_cyclic.json > /dev/null
python tools/isa_hist.py build/isa/regs_tmp.s advect_lds2_o3_kernelILi4ELb1ELi0E --json profiles/r05/isa_hist_advect_lds2_o3_k4_cyclic.json > /dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/r05/*_pmc_traffic.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items():
        if 'advect' in k:
            print(f, d['csrc_hash'], k, v.get('hbm_bytes_per_launch'), v.get('avg_ms_kernel_trace'))
PY
