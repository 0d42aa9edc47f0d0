#!/bin/bash
set -o pipefail
O=gpurun_out/r6l; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc; tail -4 $O/tests.log
python - <<'PY'
# first-touch upload of a reanalysis-sized series: torch's pageable copy against Engine.to_device (staging ring)
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
for rep in range(3):
    a = np.random.default_rng(rep).standard_normal((97, 720, 1440)).astype(np.float32)   # fresh pages every time
    b = a.copy()
    torch.cuda.synchronize(); t0 = time.perf_counter(); t = torch.from_numpy(a).to("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    s = eng.to_device(b, np.float32); torch.cuda.synchronize(); t2 = time.perf_counter()
    assert torch.equal(t, s)
    r0 = time.perf_counter(); x = t.cpu().numpy(); r1 = time.perf_counter(); y = eng.to_host(s); r2 = time.perf_counter()
    assert np.array_equal(x, y)
    print(f"402 MB first touch: torch H2D {1e3*(t1-t0):.1f} ms, staged {1e3*(t2-t1):.1f} ms | torch D2H {1e3*(r1-r0):.1f} ms, staged {1e3*(r2-r1):.1f} ms", flush=True)
PY
