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

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


def unshard_lanes(gathered, counts):
    """[world, ..., n_max] gathered slabs -> [..., total] in global lane order
    (shards are contiguous, so this is a concatenation along the lane axis)."""
    return torch.cat([gathered[r][..., :counts[r]] for r in range(len(counts))], dim=-1)
