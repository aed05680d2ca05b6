"""Multi-GPU sharding of the env batch: one process per GPU, contiguous index ranges, and the single
observation gather of BASELINE.json's north_star.

The envs are independent, so the only data-path exchange is returning the stacked observation tensor
(and, optionally, reward/done) to the rank that runs the policy, and sending that rank's actions back.
Both are plain `torch.distributed` collectives -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.  Each non-root shard rides its own xGMI link into rank 0 (gather = grouped send/recv),
so a step costs one shard transfer time, not seven (DESIGN.md "Multi-GPU").  There is never more than ONE
collective per step: when reward and done are wanted too, every shard packs [obs | reward | done] into
[count, obs_dim + 2] float32 rows -- written by the step launch itself (gaq_set_packed_rows_dev: assembled in the step kernel's LDS
buffer, no launch between the step and the collective; gaq_pack_rows_dev is the stand-alone form) -- and the rows travel together
("a 20-word row to keep it a single collective").

The reference has no counterpart (it is single-process); the only contract is that results
do not depend on the sharding: RNG streams are keyed by the GLOBAL env index (gaq_config.env_id_offset).
"""
import os


def shard_range(total, rank, world, align=1):
    """Contiguous, balanced split of range(total): the first ranks get one extra unit.  `align` > 1 splits in units
    of `align` envs (swarm worlds must not straddle shards); `total` must then be a multiple of it."""
    total, align = int(total), int(align)
    if total % align:
        raise ValueError("total_envs must be a multiple of %d (whole worlds)" % align)
    base, extra = divmod(total // align, int(world))
    first = rank * base + min(rank, extra)
    return first * align, (base + (1 if rank < extra else 0)) * align


class ShardedQuadrotorEnv(object):
    """total_envs environments split over the ranks of a torch.distributed process group.

    make_env(num_envs=, env_id_offset=, device=, **env_kwargs) builds the local shard (default:
    gym_art_amd.QuadrotorEnv); its step_dev(actions, obs, rew, done) must fill device tensors.
    """

    def __init__(self, total_envs, make_env=None, group=None, root=0, tensor_device=None, always_collective=False,
                 fused_rows=True, **env_kwargs):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.root = root
        # issue the collective even with one rank (bench.py GAQ_BENCH_FORCE_DIST=1: RCCL's gather path on a 1-GPU box)
        self._skip = (self.world == 1) and not (always_collective and dist.is_initialized())
        self.total_envs = int(total_envs)
        swarm = env_kwargs.get("swarm")
        align = int(swarm.get("agents", 8)) if swarm else 1      # a world lives on one GPU
        self._align = align
        self.first, self.count = shard_range(total_envs, self.rank, self.world, align)
        self.max_count = shard_range(total_envs, 0, self.world, align)[1]
        # every shard has max_count envs (the padded [world, max_count, ...] buffers ARE the stacked tensors).  In units of `align`:
        # 12 worlds of 8 agents over 8 ranks are 96 envs, divisible by 8, and still ragged (2, 2, 2, 2, 1, 1, 1, 1 worlds)
        self._even = self.max_count * self.world == self.total_envs
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if make_env is None:
            from .quadrotor import QuadrotorEnv as make_env
        self.env = make_env(num_envs=self.count, env_id_offset=self.first, device=local_rank, **env_kwargs)
        self.obs_dim = self.env.obs_dim
        dev = tensor_device if tensor_device is not None else torch.device("cuda", local_rank)
        self.device = dev
        f32 = torch.float32
        # shards are padded to the largest one so that every rank contributes equally sized tensors
        self._obs = torch.zeros((self.max_count, self.obs_dim), dtype=f32, device=dev)
        self._rew = torch.zeros((self.max_count,), dtype=f32, device=dev)
        self._done = torch.zeros((self.max_count,), dtype=torch.uint8, device=dev)
        self._act = torch.zeros((self.max_count, 4), dtype=f32, device=dev)
        self._rows = torch.zeros((self.max_count, self.obs_dim + 2), dtype=f32, device=dev)    # [obs | reward | done]
        self._gather_obs = None
        self._gather_rows = None
        self._act_all, self._scatter_parts = None, None
        # the packed rows are written by the step launch itself when the shard can do it (gaq_set_packed_rows_dev); a stand-in shard
        # (CPU tests) or fused_rows=False keeps the separate pack launch (pack_rows_dev)
        # (never without a collective to feed: the rows are 80 B per env-step of extra writes)
        self.fused_rows = bool(fused_rows) and hasattr(self.env, "set_packed_rows") and not self._skip
        if self.fused_rows:
            self.env.set_packed_rows(self._rows[:self.count])
        self.collectives = 0          # data-path collectives issued so far (tests: one per step)
        if self.rank == root:
            # one contiguous [world, max_count, ...] buffer per quantity: the collective writes each shard
            # straight into its slot, so the stacked tensor is a view (no per-step concatenation)
            self._obs_all = torch.zeros((self.world, self.max_count, self.obs_dim), dtype=f32, device=dev)
            self._rows_all = torch.zeros((self.world, self.max_count, self.obs_dim + 2), dtype=f32, device=dev)
            self._gather_obs = list(self._obs_all.unbind(0))
            self._gather_rows = list(self._rows_all.unbind(0))

    # -- local views -------------------------------------------------------------------------------------
    @property
    def obs(self):
        return self._obs[:self.count]

    @property
    def reward(self):
        return self._rew[:self.count]

    @property
    def done(self):
        return self._done[:self.count]

    def _stack(self, whole, parts):
        if self._even:
            return whole.reshape((self.total_envs,) + tuple(whole.shape[2:]))      # a view
        return self._torch.cat([p[:shard_range(self.total_envs, r, self.world, self._align)[1]] for r, p in enumerate(parts)], dim=0)

    # -- collectives -------------------------------------------------------------------------------------
    def gather_obs(self):
        """The north_star's single collective: every shard's obs -> rank `root`; returns the stacked
        [total_envs, obs_dim] tensor there, None elsewhere.  For evenly divisible batches the result is a view
        of the persistent gather buffer (overwritten by the next gather): clone it to keep it."""
        if self._skip:
            return self.obs
        self._dist.gather(self._obs, self._gather_obs, dst=self.root, group=self.group)
        self.collectives += 1
        return self._stack(self._obs_all, self._gather_obs) if self.rank == self.root else None

    def set_fused_rows(self, enabled):
        """Switch between the rows written by the step launch (default) and the separate pack launch (measurements)."""
        enabled = bool(enabled) and hasattr(self.env, "set_packed_rows") and not self._skip
        if hasattr(self.env, "set_packed_rows"):
            self.env.set_packed_rows(self._rows[:self.count] if enabled else None)
        self.fused_rows = enabled

    def gather_packed(self, done_as_float=False, packed=False):
        """obs, reward AND done in ONE collective: every shard packs its [count, obs_dim + 2] rows
        [obs | reward | (float) done] (gaq_pack_rows_dev) and the rows are gathered to rank `root`.  Returns
        (obs [total, obs_dim], reward [total], done [total] uint8) there -- obs and reward are views of the
        persistent gather buffer when the batch divides evenly -- and (None, None, None) elsewhere.
        `done_as_float`: hand back the rows' own 0.0 / 1.0 done column (a view) instead of converting it to uint8 --
        no extra pass over the gathered rows on rank `root`.  `packed`: the caller has made the rows already."""
        D = self.obs_dim
        if not self.fused_rows and not packed:      # otherwise the rows of the last step are already there: the step launch wrote them
            self.env.pack_rows_dev(self.obs, self.reward, self.done, self._rows[:self.count])
        if self._skip:
            rows = self._rows[:self.count]
        else:
            self._dist.gather(self._rows, self._gather_rows, dst=self.root, group=self.group)
            self.collectives += 1
            if self.rank != self.root:
                return None, None, None
            rows = self._stack(self._rows_all, self._gather_rows)
        done = rows[:, D + 1]
        return rows[:, :D], rows[:, D], (done if done_as_float else (done != 0).to(self._torch.uint8))

    def gather_reward_done(self):
        """Reward and done alone (kept for callers that already hold the observations): the same single packed
        collective as gather_packed()."""
        _, rew, done = self.gather_packed()
        return rew, done

    def scatter_actions(self, actions_global=None):
        """Rank `root` holds actions [total_envs, 4]; every rank receives its own contiguous slice."""
        if self.world == 1:
            self._act[:self.count] = actions_global
            return self._act[:self.count]
        parts = None
        if self.rank == self.root:
            # one persistent [world, max_count, 4] staging buffer (allocated on first use): every shard's slice is copied into
            # its slot, the slots ARE the scatter list -- no allocation per call
            if self._act_all is None:
                self._act_all = self._torch.zeros((self.world, self.max_count, 4), dtype=self._torch.float32, device=self.device)
                self._scatter_parts = list(self._act_all.unbind(0))
            if self._even:
                self._act_all.view(self.total_envs, 4).copy_(actions_global)
            else:
                for r in range(self.world):
                    f, c = shard_range(self.total_envs, r, self.world, self._align)
                    self._act_all[r, :c] = actions_global[f:f + c]
            parts = self._scatter_parts
        self._dist.scatter(self._act, parts, src=self.root, group=self.group)
        return self._act[:self.count]

    # -- env API -----------------------------------------------------------------------------------------
    def reset(self):
        self.env.reset_dev(self.obs)
        if self.fused_rows:          # keep rows == pack(obs, reward, done) between steps too (the step launches maintain it from here on)
            self.env.pack_rows_dev(self.obs, self.reward, self.done, self._rows[:self.count])
        return self.gather_obs()

    def step(self, actions_local, gather=True, gather_reward_done=False):
        """Step the local shard with actions_local [count, 4]; gather per the flags -- at most ONE collective:
        observations alone (`gather`), or the packed [obs | reward | done] rows (`gather_reward_done`).  Returns
        (stacked_obs or None, (reward, done) or None)."""
        self.env.step_dev(actions_local, self.obs, self.reward, self.done)
        if gather_reward_done:
            obs, rew, done = self.gather_packed()
            return (obs if gather else None), (rew, done)
        return (self.gather_obs() if gather else None), None
