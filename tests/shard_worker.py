"""One rank of the two-rank HIP shard test (tests/test_baseline_configs_gpu.py): a fresh
process that steps ITS lane shard on the GPU through the product path and takes part in
the product's one collective.  gloo stands in for RCCL (a one-GPU box cannot host two RCCL
ranks); the gathered tensors are host copies, the stepping is the HIP library."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    total, V, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abrsimulator_amd._lib import OBS_DIM
    from abrsimulator_amd.sharding import (ObsRewardGather, lane_assignment, make_slab, shard_range,
                                           unshard_lanes)
    from helpers import make_env
    from test_baseline_configs_gpu import META, _traces
    traces = _traces(16, seed=9)
    lane0, n = shard_range(total, world, rank)
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    assert len(set(counts)) == 1, "equal shards keep the gather shapes equal"
    env = make_env(dict(META, video_length=V), traces, n, auto_reset=True, lane_id_base=lane0)
    tid, off = lane_assignment(lane0, n, [len(t) for t in traces])
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    slab, obs, reward, send = make_slab(V, OBS_DIM, n, "cuda")
    env.step_random(V, seed, out=dict(obs=obs, reward=reward, done=None, actions=None))
    g = ObsRewardGather((OBS_DIM, n), (V, n), "cpu")
    go, gr = g.gather(0, send.cpu())
    g.finish()
    if rank == 0:
        np.savez(out, obs=unshard_lanes(go, counts).numpy(), reward=unshard_lanes(gr, counts).numpy(),
                 n_collectives=g.n_collectives)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
