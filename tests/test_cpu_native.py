"""bench.py's compiled CPU baseline (oracle/cpu_native.cpp: the kernel arithmetic header built for the host, OpenMP
over the batch) against the reference's golden trajectories: what it times is the reference's path."""
import numpy as np

from oracle import cpu_native as cn
from tests import golden_util as gu
from tests import gpu_util


def test_batch_driver_reproduces_hummingbird_and_crazyflie_fixtures():
    for name in ("g2_hummingbird_raw", "g3_crazyflie"):
        d = gu.load(name)
        blocks = gu.env_blocks(d)
        const = gu.sub(d, "const_")
        n = 2 * len(blocks) + 1                     # replicas + a ragged tail, two threads
        b = cn.Batch(n, const, dt=float(blocks[0]["dt"]), sim_steps=int(blocks[0]["sim_steps"]), ep_len=int(blocks[0]["ep_len"]))
        b.set_state(gpu_util.planes_from_blocks(blocks, n))
        T = max(blk["obs"].shape[0] for blk in blocks)
        obs = np.zeros((T, n, 18), np.float32); rew = np.zeros((T, n), np.float32); done = np.zeros((T, n), bool)
        for t in range(T):
            a = np.zeros((n, 4), np.float32)
            for i in range(n):
                blk = blocks[i % len(blocks)]
                if t < blk["actions"].shape[0]:
                    a[i] = blk["actions"][t]
            obs[t], rew[t], done[t] = b.step(a, threads=2)
        for k, blk in enumerate(blocks):
            Tb = blk["obs"].shape[0]
            for r in range(k, n, len(blocks)):
                assert gu.rel_err(obs[:Tb, r], blk["obs"]) <= 2e-7
                assert np.max(np.abs(rew[:Tb, r] - blk["reward"])) <= 1e-7
                assert np.array_equal(done[:Tb, r], blk["done"])
        b.close()


def test_timed_loop_runs_with_noise_and_resets():
    b = cn.Batch(4096, cn.HUMMINGBIRD, noise=1, auto_reset=1, ep_len=5)
    b.reset(2)
    steps, el = b.run(0.05, 2)
    assert steps >= 3 and el > 0
    obs, rew, done = b.step(np.zeros((4096, 4), np.float32), threads=2)
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    b.close()
