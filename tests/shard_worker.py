"""One rank of the two-rank HIP shard test (tests/test_baseline_configs_gpu.py): a fresh
process that steps ITS lane shard on the GPU through the product path and takes part in
the product's one collective -- all of it through the package's ShardedABREnv.  gloo stands in for
RCCL (a one-GPU box cannot host two RCCL ranks); the stepping is the HIP library."""
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
    import abrsimulator_amd as A
    from abrsimulator_amd.sharding import ShardedABREnv, shard_range
    from test_baseline_configs_gpu import META, _traces
    traces = _traces(16, seed=9)
    meta = dict(META, video_length=V)
    # the package's N > 1 composition: shard range, lane_id_base, the global lane map, slabs, the one collective
    sh = ShardedABREnv(A.MPD(meta["video_length"], meta["chunk_length"], meta["max_buffer"], meta["start_up_length"],
                             A.Chunk(meta["ladder"])), A.QOEMetric(*meta["weights"]),
                       A.NetworkInfo(meta["interval"], [np.asarray(t, np.float64) for t in traces]),
                       total_lanes=total, fuse=V, device="cuda", auto_reset=True)
    assert (sh.lane0, sh.n_lanes) == shard_range(total, world, rank) and len(set(sh.counts)) == 1
    sh.reset()
    st = sh.step_random(V, seed)          # the HIP launch of this rank's shard + the all-gather (gloo: via host copies)
    sh.finish()
    go, gr = st.unsharded()
    if rank == 0:
        np.savez(out, obs=go.cpu().numpy(), reward=gr.cpu().numpy(), n_collectives=sh.n_collectives)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
