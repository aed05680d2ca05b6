"""N > 1 path on CPU: two gloo ranks, a deterministic stand-in for the local shard (the real one needs a
GPU), checking the contiguous partition, the padded gather order and the action scatter."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partition():
    from gym_art_amd.sharding import shard_range
    for total, world in ((1 << 20, 8), (1000, 8), (7, 8), (65536, 3), (5, 2)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(1 << 20, 3, 8) == (3 * 131072, 131072)         # BASELINE config 4: 131 072 envs per GPU
    # swarm worlds (config 5) never straddle shards: the split is in units of `align` envs
    for total, world, align in ((8 * 13, 4, 8), (1 << 20, 8, 8), (64 * 7, 3, 64)):
        spans = [shard_range(total, r, world, align) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        assert all(f % align == 0 and c % align == 0 for f, c in spans)
        assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
    import pytest
    with pytest.raises(ValueError):
        shard_range(100, 0, 4, align=8)


class FakeShard(object):
    """obs[i, k] = 1000 * global_index + k + step; reward = global index; done = (global_index + step) % 2."""
    obs_dim = 18

    def __init__(self, num_envs, env_id_offset, device, **kw):
        self.num_envs, self.off, self.t = num_envs, env_id_offset, 0
        self.kw = kw

    def _fill(self, obs):
        import torch
        g = torch.arange(self.off, self.off + self.num_envs, dtype=torch.float32)
        obs.copy_(1000.0 * g[:, None] + torch.arange(18, dtype=torch.float32)[None] + self.t)
        return g

    def reset_dev(self, obs):
        self._fill(obs)

    def step_dev(self, actions, obs, rew, done):
        self.t += 1
        g = self._fill(obs)
        rew.copy_(g + actions.sum(1) + 0.123456789)      # not exactly representable sums: the packed row must carry the bits
        done.copy_(((g.long() + self.t) % 2).to(done.dtype))

    def pack_rows_dev(self, obs, rew, done, rows):
        """stand-in for gaq_pack_rows_dev (the real one is a HIP launch): [obs | reward | float(done)]"""
        rows[:, :18] = obs
        rows[:, 18] = rew
        rows[:, 19] = done.float()


class FusedShard(FakeShard):
    """... and a shard that writes the packed rows ITSELF with every step, like the F_ROWS twins of the real kernels do once a row
    buffer is registered (gaq_set_packed_rows_dev): the sharding layer must then NOT call pack_rows_dev between step and gather."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.rows, self.pack_calls = None, 0

    def set_packed_rows(self, rows):
        self.rows = rows

    def step_dev(self, actions, obs, rew, done):
        super().step_dev(actions, obs, rew, done)
        if self.rows is not None:
            FakeShard.pack_rows_dev(self, obs, rew, done, self.rows)

    def pack_rows_dev(self, obs, rew, done, rows):
        self.pack_calls += 1
        super().pack_rows_dev(obs, rew, done, rows)


def _worker(rank, world, total, port, out, agents=0):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gym_art_amd.sharding import ShardedQuadrotorEnv, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = {"ep_time": 5}
        if agents:
            kw["swarm"] = {"agents": agents}          # whole worlds per shard: the split is in units of `agents` envs
        env = ShardedQuadrotorEnv(total, make_env=FakeShard, tensor_device=torch.device("cpu"), **kw)
        assert (env.first, env.count) == shard_range(total, rank, world, max(agents, 1)) and env.env.kw == kw
        assert env.count % max(agents, 1) == 0
        obs0 = env.reset()
        obs0 = None if obs0 is None else obs0.clone()      # the stacked tensor is a view of the gather buffer
        glob_actions = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4) if rank == 0 else None
        act = env.scatter_actions(glob_actions)
        expect = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4)[env.first:env.first + env.count]
        assert torch.equal(act, expect)
        before = env.collectives
        obs1, (rew, done) = env.step(act, gather=True, gather_reward_done=True)
        assert env.collectives == before + 1               # obs + reward + done travel in ONE collective
        local = (env.obs.clone(), env.reward.clone(), env.done.clone())
        if rank == 0:
            # the packed rows carry every shard's obs / reward / done bit for bit
            assert obs1.dtype == torch.float32 and rew.dtype == torch.float32 and done.dtype == torch.uint8
            assert torch.equal(obs1[:env.count], local[0]) and torch.equal(rew[:env.count], local[1])
            assert torch.equal(done[:env.count], local[2])
        obs_only, none = env.step(act, gather=True)         # obs alone: also one collective, reward/done stay local
        assert none is None and env.collectives == before + 2
        if rank == 0:
            g = torch.arange(total, dtype=torch.float32)
            assert obs0.shape == (total, 18) and torch.equal(obs0, 1000.0 * g[:, None] + torch.arange(18.0)[None])
            assert torch.equal(obs1, 1000.0 * g[:, None] + torch.arange(18.0)[None] + 1)
            assert torch.equal(rew, g + torch.arange(total * 4, dtype=torch.float32).reshape(total, 4).sum(1) + 0.123456789)
            assert torch.equal(done.long(), (g.long() + 1) % 2)
            assert torch.equal(obs_only, 1000.0 * g[:, None] + torch.arange(18.0)[None] + 2)
        else:
            assert obs0 is None and obs1 is None and rew is None
        # the action scatter reuses ONE persistent staging buffer on the root (no allocation per call) and can be repeated
        act2 = env.scatter_actions(glob_actions * 2 if rank == 0 else None)
        assert torch.equal(act2, expect * 2)
        if rank == 0:
            buf = env._act_all
            env.scatter_actions(glob_actions)
            assert env._act_all is buf
        else:
            env.scatter_actions(None)
        # the same with a shard that writes its packed rows itself: registered at construction, kept consistent by reset(), and
        # no pack call between a step and its gather; switching the fusion off brings the pack call back
        fenv = ShardedQuadrotorEnv(total, make_env=FusedShard, tensor_device=torch.device("cpu"), **({"swarm": kw["swarm"]} if agents else {}))
        assert fenv.fused_rows and fenv.env.rows is not None
        fenv.reset()
        packs = fenv.env.pack_calls                           # (reset packs once: rows == pack(obs, reward, done) from the start)
        o, r, d = fenv.gather_packed()
        if rank == 0:
            g = torch.arange(total, dtype=torch.float32)
            assert torch.equal(o, 1000.0 * g[:, None] + torch.arange(18.0)[None])
        fo, (fr, fd) = fenv.step(act, gather=True, gather_reward_done=True)
        assert fenv.env.pack_calls == packs
        if rank == 0:
            assert torch.equal(fo, obs1) and torch.equal(fr, rew) and torch.equal(fd, done)
        fenv.set_fused_rows(False)
        assert fenv.env.rows is None
        fenv.step(act, gather=True, gather_reward_done=True)
        assert fenv.env.pack_calls == packs + 1
        out.put((rank, "ok"))
    except Exception as e:      # surface the failure in the parent
        out.put((rank, "FAIL %r" % (e,)))
    finally:
        dist.destroy_process_group()


def _run_ranks(world, total, agents=0):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total % 997 + world
    procs = [ctx.Process(target=_worker, args=(r, world, total, port, out, agents)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {r: "ok" for r in range(world)}, res


@pytest.mark.parametrize("total", [64, 37])      # even and ragged split
def test_two_rank_gather_and_scatter(total):
    _run_ranks(2, total)


@pytest.mark.parametrize("total", [1024, 1001])  # config 4's rank count (BASELINE.json: 8 GPUs), even and ragged split
def test_eight_rank_gather_and_scatter(total):
    """The driver's N = 8 run is the only one this code gets on eight ranks: the same worker -- shard ranges, ONE packed gather per step,
    action scatter, fused rows -- with world_size 8 over gloo."""
    _run_ranks(8, total)


@pytest.mark.parametrize("world,worlds", [(2, 3), (8, 12)])
def test_ragged_split_of_swarm_worlds_whose_env_count_divides_evenly(world, worlds):
    """12 worlds of 8 agents over 8 ranks are 96 envs -- divisible by 8 -- in shards of 16, 16, 16, 16, 8, 8, 8, 8: the
    "evenly divisible" fast paths of the gather stack and the action scatter must go by WORLDS per rank, not by envs per rank (a view
    of the padded [world, max_count, ...] buffer as [total_envs, ...] raised on the root while the other ranks blocked in the scatter)."""
    _run_ranks(world, worlds * 8, agents=8)
