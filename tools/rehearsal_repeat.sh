#!/bin/bash
# usage: tools/rehearsal_repeat.sh <n> ; the two-process c4 rehearsal (both ranks on GPU 0, gloo) n times, stops at the first failed halo check
N=${1:-10}
mkdir -p gpurun_out/rehearsal
for i in $(seq 1 $N); do
  LCS_BENCH_BACKEND=gloo LCS_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 1 --warmup 1 --workload c4 --seeds 512 --nt 9 > gpurun_out/rehearsal/run_$i.json 2> gpurun_out/rehearsal/run_$i.err || { echo "run $i exited non-zero"; tail -n 5 gpurun_out/rehearsal/run_$i.err; exit 1; }
  ok=$(python -c "import json,sys; d=json.loads(open('gpurun_out/rehearsal/run_$i.json').read().strip().splitlines()[-1]); print(d['halo_check']['timed_path_ok'])")
  echo "run $i: timed_path_ok $ok"
  if [ "$ok" != "True" ]; then python -c "import json; d=json.loads(open('gpurun_out/rehearsal/run_$i.json').read().strip().splitlines()[-1]); print(json.dumps(d['halo_check'], indent=1))"; rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2; exit 0; fi
done
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -2
echo "all $N clean"
