"""Worker of tests/test_bench_harness_cpu.py: two gloo ranks go through bench.py's harness -- process group with a time-out,
phases under the watchdog, the progressive JSON line, guarded blocks -- with a stand-in for the GPU work.  Rank 1 leaves the
run in the way MODE says, right after the headline:
   die     os._exit(17): a rank that crashed
   stall   sleeps for ever: a rank that hangs (a collective that never completes)
   raise   a later block raises on every rank: the run goes on and ends with exit code 0
"""
import os
import sys
import time
from datetime import timedelta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

mode, dist_timeout, phase_timeout = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
wd = bench.Watchdog(rank, world, default_bound=phase_timeout, poll=0.05)
em = bench.Emitter(rank)
with wd.phase("init_process_group", dist_timeout + 10):
    dist.init_process_group("gloo", timeout=timedelta(seconds=dist_timeout))


def barrier():
    dist.barrier()


def timed_region(n):
    barrier()
    t0 = time.perf_counter()
    x = torch.zeros(1024)
    for _ in range(n):
        x = x + 1.0                       # the "launches"
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


with wd.phase("timed_region[0]"):
    el = timed_region(20)
em.update(metric="env_steps_per_sec", value=20 * 1024 * world / el, unit="env-steps/s", n_gpus=world, steps=20, warmup=0,
          config={"workload": "standin", "overrides": bench.overrides_in_force()})
em.emit()                                  # the headline is on stdout from here on

if mode in ("die", "stall") and rank == 1:
    if mode == "die":
        os._exit(17)
    time.sleep(3600)


def block_a():
    if mode == "raise":
        raise RuntimeError("block a broke")
    return {"value": timed_region(10)}     # rank 0 waits here for a peer that is gone or silent


def block_b():
    return {"value": timed_region(5)}


fatal = mode != "raise"                    # bench.py: fatal = more than one rank; "raise" plays the one-rank case, where the run goes on
bench.guarded(em, wd, "block_a", block_a, fatal=fatal)
bench.guarded(em, wd, "block_b", block_b, fatal=fatal)
with wd.phase("destroy_process_group", 10):
    dist.destroy_process_group()
wd.stop()
