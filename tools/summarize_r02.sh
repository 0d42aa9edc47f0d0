#!/bin/bash
# Reduce gpurun_out/r02_{c3_o1,c3_o3,c2} (tools/prof_all.sh) to the summaries under profiles/r02/, stamped with csrc_hash.
set -e
for o in 1 3; do
  W="{\"workload\":\"c3\",\"seeds\":4096,\"nt\":97,\"order\":$o,\"K\":4,\"dtype\":\"f32\"}"
  python profiles/summarize.py gpurun_out/r02_c3_o$o profiles/r02/c3_o$o "$W" > /dev/null
  python profiles/summarize_sq.py gpurun_out/r02_c3_o$o profiles/r02/c3_o$o "$W" 96 > /dev/null
  cp gpurun_out/r02_c3_o$o/bench_stdout.json profiles/r02/c3_o${o}_bench_stdout.json
done
W='{"workload":"c2","order":1,"K":4,"dtype":"f64","fuse_levels":false}'
python profiles/summarize.py gpurun_out/r02_c2 profiles/r02/c2 "$W" > /dev/null
python profiles/summarize_sq.py gpurun_out/r02_c2 profiles/r02/c2 "$W" 200 > /dev/null
cp gpurun_out/r02_c2/bench_stdout.json profiles/r02/c2_bench_stdout.json
tools/regs.sh > /dev/null
python tools/isa_hist.py build/isa/regs_tmp.s advect_lds2_kernelILi4ELb1E --json profiles/r02/isa_hist_advect_lds2_k4_cyclic.json > /dev/null
python tools/isa_hist.py build/isa/regs_tmp.s advect_lds_kernelILi3ELi4ELb1E --json profiles/r02/isa_hist_advect_lds_o3_k4_cyclic.json > /dev/null
python tools/isa_hist.py build/isa/regs_tmp.s advect_lds_kernelILi1ELi4ELb1E --json profiles/r02/isa_hist_advect_lds_o1_k4_cyclic.json > /dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/r02/*_pmc_traffic.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items():
        if 'advect' in k:
            print(f, d['csrc_hash'], k, v.get('hbm_bytes_per_launch'), v.get('avg_ms_kernel_trace'))
PY
