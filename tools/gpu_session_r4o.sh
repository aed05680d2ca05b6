#!/bin/bash
# round 4, session O: the small-batch host path of gaq_step without copies (mapped host memory): the whole GPU suite (its NumPy step() calls
# all take that path), then the single-env drop-in loop and N = 64 ... 8192 with and without (GAQ_ZERO_COPY=0)
set -o pipefail
O=gpurun_out/${1:-r4o}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputest.log
tail -8 $O/gputest.log
for z in 1 0 1 0; do
GAQ_ZERO_COPY=$z timeout -k 10 300 python - <<PY | tee -a $O/host_path.txt
import time, numpy as np
from gym_art_amd import QuadrotorEnv
for n, kw in ((1, {}), (1, dict(raw_control=False)), (64, {}), (1024, {}), (8192, {})):
    env = QuadrotorEnv(num_envs=n, ep_time=5, seed=0, **kw)
    env.reset()
    a = np.random.RandomState(0).uniform(-1, 1, (n, 4)).astype(np.float32)
    a = a[0] if n == 1 else a
    for _ in range(200): env.step(a)
    t0 = time.perf_counter()
    for _ in range(3000): env.step(a)
    dt = (time.perf_counter() - t0) / 3000
    print("GAQ_ZERO_COPY=$z  n=%5d %-22s %7.2f us per step()  %.3e env-steps/s" % (n, kw, dt * 1e6, n / dt))
    env.close()
PY
done
exit 0
