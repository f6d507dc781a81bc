"""Multi-GPU layout of a batched environment: contiguous lane shards, one process
per GPU, no exchange inside a step, ONE collective per launch -- an all-gather of
the (obs, reward) slabs (RCCL over xGMI on the GPU box; gloo in the CPU tests).

The reference has no distributed code at all; lanes are
independent (no cross-lane term anywhere in Simulator.py:135-208 or mpc.py:120-162),
so sharding is a partition of the lane index space.  The built-in random policy is
counter-based on the GLOBAL lane id (abr_env_set_lane_id_base), so an N-shard run
reproduces the 1-shard run lane for lane.
"""
from typing import Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total_lanes: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first global lane, number of lanes) of `rank`; shards differ by at most one lane."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_lanes), int(world_size))
    n = base + (1 if rank < rem else 0)
    lane0 = rank * base + min(rank, rem)
    return lane0, n


def lane_assignment(lane0: int, n: int, trace_lens: Sequence[int], xcd_groups: int = 0):
    """Deterministic global-lane -> (trace_id, start_offset) map used by bench.py:
    lane i reads trace i % n_traces from offset (i * 2654435761 mod 2^32) % len.
    xcd_groups = 8 makes the map XCD-aware: workgroup w = lane // 64 runs on XCD group
    w % 8 (workgroups are dealt round-robin over the 8 XCDs), and that group only reads
    traces t with t % 8 == w % 8, so each XCD's L2 holds 1/8 of the trace table."""
    i = np.arange(lane0, lane0 + n, dtype=np.uint64)
    lens = np.asarray(trace_lens, np.uint64)
    if xcd_groups and len(lens) % xcd_groups == 0:
        g = np.uint64(xcd_groups)
        per = np.uint64(len(lens) // xcd_groups)
        # GLOBAL lane ids, so an N-shard run keeps the 1-shard lane -> trace map; the
        # workgroup of a shard-local lane equals the global one (mod 8) whenever lane0 is a
        # multiple of 512, which equal shards of the bench sizes are
        w = i // np.uint64(64)                         # workgroup (one wave of 64 lanes)
        q = (w // g) * np.uint64(64) + i % np.uint64(64)     # lane's rank inside its XCD group
        tid = ((q % per) * g + (w % g)).astype(np.int32)
    else:
        tid = (i % np.uint64(len(lens))).astype(np.int32)
    off = ((i * np.uint64(2654435761)) % np.uint64(2 ** 32) % lens[tid]).astype(np.int32)
    return tid, off


def make_slab(n_steps: int, obs_dim: int, n_lanes: int, device):
    """One contiguous float32 slab laid out [obs: n_steps x obs_dim x N | reward: n_steps x N].
    The kernel writes obs and reward through the two views; `send` -- the LAST step's
    observation followed by all rewards -- is contiguous inside the slab, so the collective
    ships it without a packing copy.  Returns (slab, obs, reward, send)."""
    n_obs = n_steps * obs_dim * n_lanes
    slab = torch.empty(n_obs + n_steps * n_lanes, dtype=torch.float32, device=device)
    obs = slab[:n_obs].view(n_steps, obs_dim, n_lanes)
    reward = slab[n_obs:].view(n_steps, n_lanes)
    send = slab[n_obs - obs_dim * n_lanes:]
    return slab, obs, reward, send


class ObsRewardGather:
    """THE one collective of the path: a double-buffered all-gather of each rank's packed
    (obs, reward) slab -- a single all_gather_into_tensor per launch (RCCL over xGMI on the
    GPU box; gloo in the CPU tests) -- issued on a side stream so that it overlaps the next
    launch when the tensors live on a GPU.

    obs_shape / reward_shape describe one rank's part, e.g. (OBS_DIM, N) and (F, N): the
    sent slab is obs.numel() + reward.numel() float32 values, obs first (make_slab)."""

    def __init__(self, obs_shape, reward_shape, device, group=None, n_buffers=2):
        self.group = group
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.obs_shape, self.reward_shape = tuple(obs_shape), tuple(reward_shape)
        self.n_obs = int(np.prod(self.obs_shape))
        self.n_rew = int(np.prod(self.reward_shape))
        self.numel = self.n_obs + self.n_rew
        self.out = [torch.empty(self.world * self.numel, dtype=torch.float32, device=device)
                    for _ in range(n_buffers)]
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.pending = [None] * n_buffers
        self.n_collectives = 0

    def wait_free(self, b):
        """Before the producer overwrites source slab b again."""
        if self.cuda and self.pending[b] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.pending[b])

    def gather(self, b, send):
        """Enqueue the all-gather of this rank's packed slab into buffer b.  Returns
        (obs [world, *obs_shape], reward [world, *reward_shape]) views of the result."""
        if send.numel() != self.numel or send.dtype != torch.float32 or not send.is_contiguous():
            raise ValueError(f"send slab must be {self.numel} contiguous float32 values")
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                dist.all_gather_into_tensor(self.out[b], send, group=self.group)
                fin = torch.cuda.Event()
                fin.record()
            self.pending[b] = fin
        else:
            dist.all_gather_into_tensor(self.out[b], send, group=self.group)
        self.n_collectives += 1
        return self.split(self.out[b])

    def split(self, gathered):
        g = gathered.view(self.world, self.numel)
        return (g[:, :self.n_obs].reshape((self.world,) + self.obs_shape),
                g[:, self.n_obs:].reshape((self.world,) + self.reward_shape))

    def wait_done(self, b):
        """The caller's stream waits until the gather into buffer b has finished (its result may then be read)."""
        if self.cuda and self.pending[b] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.pending[b])

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


class ShardStep:
    """What one launch of a ShardedABREnv returns: `local` -- this rank's outputs (dict obs [F, OBS_DIM, n], reward [F, n],
    done [F, n], actions [F, n] or None; views of the launch's slab, valid until the launch after next reuses it) -- and the
    handle to the ONE collective of the launch: gathered() -> (obs [world, OBS_DIM, n_max], reward [world, F, n_max]) of
    every rank's final observation and rewards once the all-gather has finished; unsharded() the same in global lane order.
    LIFETIME of what gathered() / unsharded() return: views of (or a concatenation made from) the double-buffered receive
    buffer of the collective -- valid until the launch AFTER NEXT enqueues its all-gather into the same buffer on the side
    stream; clone() what has to live longer, and read it on the stream gathered() was called on (that stream has waited for
    the collective; another stream has not)."""

    def __init__(self, owner, buf, local, views):
        self._owner, self._buf, self.local, self._views = owner, buf, local, views

    def gathered(self):
        if self._views is None:
            raise RuntimeError("this launch was not gathered (gather=False, or a launch shape other than `fuse`)")
        self._owner._gather.wait_done(self._buf)
        return self._views

    def unsharded(self):
        go, gr = self.gathered()
        return unshard_lanes(go, self._owner.counts), unshard_lanes(gr, self._owner.counts)


class ShardedABREnv:
    """One rank's shard of ONE batched environment of `total_lanes` lanes spread over the ranks of a process group
    (one process per GPU).  Owns everything the N > 1 composition consists of: the shard's lane range
    (shard_range), the global lane ids of the counter-based policy (lane_id_base), the global lane -> (trace, offset)
    map (lane_assignment), two slabs the kernel writes (obs, reward) into, and THE one collective of the path -- a
    single all_gather_into_tensor of the packed [final observation | rewards] per launch (ObsRewardGather: RCCL over
    xGMI on the GPU box, gloo in the CPU tests), issued on a side stream and double-buffered so that it overlaps the next
    launch.  An N-rank run reproduces the 1-rank run lane for lane.

        env = ShardedABREnv(mpd, qoe, net, total_lanes=1048576, fuse=48)       # rank / world from torch.distributed
        env.reset()
        st = env.step_random(48, seed)        # this rank's launch; the all-gather is in flight
        st2 = env.step_random(48, seed)       # overlaps it
        obs, reward = st.gathered()           # [world, 8, n_max], [world, 48, n_max]
        env.finish()

    lanes_per_rank: weak scaling instead -- every rank gets that many lanes (total = world * lanes_per_rank).
    `env`: the per-rank stepper (default: a BatchedABREnv on `device`); anything with n_lanes, reset(trace_id, offset),
    step_random / step_script(out=) works -- the gloo tests put a CPU stand-in there."""

    def __init__(self, mpd, qoe_metric, network_info, total_lanes=None, lanes_per_rank=None, fuse=48, device="cuda",
                 group=None, rank=None, world=None, gather=True, obs_dim=8, env=None, **env_kw):
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
        self.rank, self.world, self.group = int(rank), int(world), group
        if (total_lanes is None) == (lanes_per_rank is None):
            raise ValueError("give total_lanes (strong scaling) or lanes_per_rank (weak scaling)")
        if lanes_per_rank is not None:
            total_lanes = int(lanes_per_rank) * self.world
        self.total_lanes = int(total_lanes)
        self.lane0, self.n_lanes = shard_range(self.total_lanes, self.world, self.rank)
        self.counts = [shard_range(self.total_lanes, self.world, r)[1] for r in range(self.world)]
        self.n_max = max(self.counts)
        self.fuse, self.obs_dim = int(fuse), int(obs_dim)
        self.device = torch.device(device)
        self.network_info = network_info
        if env is None:
            from .env import BatchedABREnv
            env = BatchedABREnv(mpd, qoe_metric, network_info, self.n_lanes, device=self.device,
                                lane_id_base=self.lane0, **env_kw)
        self.env = env
        # the collective runs whenever a process group exists (a ONE-rank group too: the same code path as N ranks)
        self._do_gather = bool(gather) and dist.is_available() and dist.is_initialized()
        F, D, n = self.fuse, self.obs_dim, self.n_lanes
        self._slabs = [make_slab(F, D, n, self.device) for _ in range(2)]
        self._done = [torch.empty(F, n, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._acts = [None, None]
        self._outs = [self._bind(dict(obs=o, reward=r, done=self._done[b], actions=None))
                      for b, (_, o, r, _) in enumerate(self._slabs)]
        self._gather = ObsRewardGather((D, self.n_max), (F, self.n_max), self.device, group=group) if self._do_gather else None
        # uneven shards: the collective needs equal parts, so a smaller shard sends through a zero-padded staging copy
        self._stage = ([torch.zeros(D + F, self.n_max, dtype=torch.float32, device=self.device) for _ in range(2)]
                       if self._do_gather and n != self.n_max else None)
        self._it = 0

    def _bind(self, out):
        return self.env.bind_out(out) if hasattr(self.env, "bind_out") else out

    # ---- the global lane -> (trace, offset) map ----
    def lane_map(self, xcd_groups=0):
        bw = self.network_info.bandwidths
        lens = [len(t) for t in (bw if hasattr(bw[0], "__len__") else [bw])]
        return lane_assignment(self.lane0, self.n_lanes, lens, xcd_groups=xcd_groups)

    def reset(self, trace_id=None, start_offset=None, mask=None):
        """Default: the deterministic global-lane map (lane_assignment); or this shard's own trace ids / offsets.
        As BatchedABREnv.reset(): no host synchronisation; a lane with a trace id out of range or a negative offset is frozen
        on the device with ABR_DONE_BADARG (visible in self.env.done_after_reset(), or in `done` after the next launch) while
        its returned observation looks like a fresh lane's."""
        if trace_id is None:
            trace_id, start_offset = self.lane_map()
        tid = torch.as_tensor(trace_id)
        off = None if start_offset is None else torch.as_tensor(start_offset)
        return self.env.reset(tid, off) if mask is None else self.env.reset(tid, off, mask)

    # ---- launches ----
    def _launch(self, n_steps, call, events=None):
        """`call(out)` runs the launch into the dict `out`; returns the ShardStep with the collective in flight.
        events: an optional (start, end) pair of HIP events recorded right around the kernel launch itself."""
        if n_steps != self.fuse:                       # an odd launch shape: local outputs only
            if events:
                events[0].record()
            local = call(None)
            if events:
                events[1].record()
            return ShardStep(self, None, local, None)
        b = self._it & 1
        self._it += 1
        if self._gather is not None:
            self._gather.wait_free(b)                  # slab b's previous gather has read it
        if events:
            events[0].record()
        local = call(self._outs[b])
        if events:
            events[1].record()
        views = None
        if self._gather is not None:
            send = self._slabs[b][3]
            if self._stage is not None:
                st = self._stage[b]
                st[:, :self.n_lanes].copy_(send.view(self.obs_dim + self.fuse, self.n_lanes))
                send = st.view(-1)
            views = self._gather.gather(b, send)
        return ShardStep(self, b, local, views)

    def step_random(self, n_steps, seed, events=None):
        """n_steps fused decisions per lane under the built-in counter-based policy (global lane ids)."""
        return self._launch(int(n_steps), lambda out: self.env.step_random(int(n_steps), seed, out=out), events)

    def step_script(self, actions):
        """actions int32 [n_steps, n_lanes] of THIS shard's lanes."""
        n = int(actions.shape[0])
        return self._launch(n, lambda out: self.env.step_script(actions, out=out))

    def step_mpc(self, controller, n_steps):
        """n_steps decisions taken by `controller` on this shard's own state (actions are computed where the lane lives;
        only (obs, reward) is gathered)."""
        n = int(n_steps)

        def call(out):
            if out is not None and out.get("actions") is None:
                out = dict(out)
                out["actions"] = torch.empty(n, self.n_lanes, dtype=torch.int32, device=self.device)
            return self.env.step_mpc(controller, n, out=out)
        return self._launch(n, call)

    @property
    def n_collectives(self):
        return self._gather.n_collectives if self._gather is not None else 0

    def finish(self):
        """The caller's stream waits for every collective in flight."""
        if self._gather is not None:
            self._gather.finish()


def unshard_lanes(gathered, counts):
    """[world, ..., n_max] gathered slabs -> [..., total] in global lane order
    (shards are contiguous, so this is a concatenation along the lane axis)."""
    return torch.cat([gathered[r][..., :counts[r]] for r in range(len(counts))], dim=-1)
