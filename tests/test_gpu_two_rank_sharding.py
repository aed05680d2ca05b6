"""The N > 1 path with the REAL HIP shards: two ranks (gloo -- RCCL cannot put two ranks on one device) share the one
GPU of the test box, each owning a contiguous env range.  The stacked observation gathered on rank 0 must equal,
bit for bit, what one process computes for the whole batch: RNG streams are keyed by the global env index
(gaq_config.env_id_offset), so results do not depend on the sharding."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(dynamics_params="DefaultQuad", ep_time=0.1, seed=123)      # ep_len 10: auto-resets inside the run, noise on
STEPS = 25


def _actions(total, t):
    rng = np.random.RandomState(1000 + t)
    return rng.uniform(-1, 1, size=(total, 4)).astype(np.float32)


def _worker(rank, world, total, port, path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gym_art_amd.sharding import ShardedQuadrotorEnv
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        env = ShardedQuadrotorEnv(total, tensor_device=dev, **KW)
        frames = []
        obs = env.reset()
        if rank == 0:
            frames.append(obs.cpu().numpy().copy())
        for t in range(STEPS):
            a = torch.tensor(_actions(total, t)[env.first:env.first + env.count], device=dev)
            obs, (rew, done) = env.step(a, gather=True, gather_reward_done=True)
            if rank == 0:
                frames.append(obs.cpu().numpy().copy())
                frames.append(rew.cpu().numpy().astype(np.float32)[:, None].repeat(18, 1))
                frames.append(done.cpu().numpy().astype(np.float32)[:, None].repeat(18, 1))
        if rank == 0:
            np.save(path, np.stack(frames))
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, "FAIL %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(4096, 2), (1000, 2), (1000, 3), (8 * 77, 4)])   # even, ragged, 3 and 4 ranks
def test_two_ranks_equal_one_process(total, world, tmp_path):
    import torch
    import torch.multiprocessing as mp
    from gym_art_amd import QuadrotorEnv
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + total % 97 + 100 * world
    path = str(tmp_path / "stacked.npy")
    procs = [ctx.Process(target=_worker, args=(r, world, total, port, path, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {r: "ok" for r in range(world)}, res
    sharded = np.load(path)
    # the same batch in one process
    env = QuadrotorEnv(num_envs=total, **KW)
    dev = torch.device("cuda", 0)
    obs = torch.empty((total, 18), device=dev); rew = torch.empty(total, device=dev); done = torch.empty(total, dtype=torch.uint8, device=dev)
    env.reset_dev(obs)
    torch.cuda.synchronize()
    frames = [obs.cpu().numpy().copy()]
    for t in range(STEPS):
        env.step_dev(torch.tensor(_actions(total, t), device=dev), obs, rew, done)
        torch.cuda.synchronize()
        frames.append(obs.cpu().numpy().copy())
        frames.append(rew.cpu().numpy()[:, None].repeat(18, 1))
        frames.append(done.cpu().numpy().astype(np.float32)[:, None].repeat(18, 1))
    single = np.stack(frames)
    assert sharded.shape == single.shape
    assert np.array_equal(sharded, single)
    assert single[3::3].sum() > 0          # some episodes ended (and were re-drawn identically) inside the run
