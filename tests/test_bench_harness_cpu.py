"""bench.py's harness without a GPU: the first multi-rank contact must be diagnosable.  Two gloo ranks run
tests/bench_harness_worker.py, which drives bench.Watchdog / bench.Emitter / bench.guarded exactly as bench.py's Run does; one
rank crashes or stalls right after the headline.  Rank 0 must exit non-zero within the bound, the headline line must already be
on its stdout, and -- when the watchdog ended it -- stderr must say which rank was in which phase.  Also: override bookkeeping."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "bench_harness_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, dist_timeout, phase_timeout, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(dist_timeout), str(phase_timeout)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


def _finish(procs, limit):
    t0 = time.monotonic()
    outs = []
    try:
        for p in procs[:1]:
            outs.append(p.communicate(timeout=limit))
    finally:
        for p in procs[1:]:
            p.kill()                       # the stalled / dead rank: this test's own child, by PID
            outs.append(p.communicate())
    return outs, time.monotonic() - t0


def _lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("mode", ["die", "stall"])
def test_a_lost_rank_ends_rank0_within_the_bound_with_the_headline_on_stdout(mode):
    # process-group time-out 60 s, phase bound 4 s: the watchdog is what must end a stalled run here
    procs = _launch(mode, dist_timeout=60.0, phase_timeout=4.0)
    (out0, err0), took = _finish(procs, limit=60.0)[0][0], None
    rc = procs[0].returncode
    assert rc not in (0, None), (rc, out0[-500:], err0[-1500:])
    lines = _lines(out0)
    assert len(lines) >= 1 and lines[0]["metric"] == "env_steps_per_sec" and lines[0]["value"] > 0, out0[-500:]
    assert lines[-1]["value"] == lines[0]["value"]                       # whatever was printed later still carries the headline
    if rc == 124:                                                        # ended by the watchdog: it says who was where
        assert "bench.py watchdog" in err0
        msg = json.loads(err0[err0.index("bench.py watchdog: ") + len("bench.py watchdog: "):].splitlines()[0])
        assert msg["error"] == "timeout" and msg["rank"] == 0 and msg["phase"] == "block_a" and msg["world"] == 2
        assert "timed_region[0]" in msg["completed_phases"]
    else:                                                                # gloo noticed the dead peer first: the block's error is
        assert mode == "die" and rc == 3, (rc, err0[-1500:])             # in the line, and the rank says so and stops
        assert "error" in lines[-1]["block_a"] and "block `block_a` failed" in err0 and "block_b" not in lines[-1]
    if mode == "stall":
        assert rc == 124


def test_the_bound_holds():
    """... and it is the bound that ends it: a 3 s phase bound ends rank 0 in well under the 60 s of the process group."""
    t0 = time.monotonic()
    procs = _launch("stall", dist_timeout=60.0, phase_timeout=3.0)
    _finish(procs, limit=40.0)
    assert procs[0].returncode == 124 and time.monotonic() - t0 < 25.0


def test_a_block_that_raises_is_recorded_and_the_run_goes_on():
    procs = _launch("raise", dist_timeout=30.0, phase_timeout=20.0, extra_env={"ABR_BENCH_CORES": "3"})
    outs = [p.communicate(timeout=90.0) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-800:] for o in outs]
    lines = _lines(outs[0][0])
    assert len(lines) == 3                                               # headline, + block_a, + block_b: the line grows
    assert "block_a" not in lines[0] and "error" in lines[1]["block_a"] and "block a broke" in lines[1]["block_a"]["error"]
    assert lines[2]["block_b"]["value"] > 0 and lines[2]["block_a"] == lines[1]["block_a"] and lines[2]["value"] == lines[0]["value"]
    assert lines[0]["config"]["overrides"] == {"ABR_BENCH_CORES": "3"}   # an override in force is on record
    assert not _lines(outs[1][0])                                        # only rank 0 prints


def test_workload_overrides_need_the_flag():
    import bench
    assert bench.check_overrides(False, {"ABR_BENCH_CORES": "4", "ABR_HIP_LIB": "x.so", "HOME": "/"}) == \
        {"ABR_BENCH_CORES": "4", "ABR_HIP_LIB": "x.so"}
    for k in bench.WORKLOAD_OVERRIDES:
        with pytest.raises(SystemExit, match="allow-overrides"):
            bench.check_overrides(False, {k: "1"})
        assert bench.check_overrides(True, {k: "1"}) == {k: "1"}
    assert bench.check_overrides(False, {"ABR_BENCH_INTERVAL": ""}) == {}          # unset-by-empty does not count
    a = bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert not a.allow_overrides and a.dist_timeout == 90.0 and a.phase_timeout == 150.0
