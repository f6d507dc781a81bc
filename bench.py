#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
  BASELINE.json's 1/2/4/8-GPU scaling curve on 1 048 576 lanes is measured in the same process and
  reported under "strong_1048576" (or run only that job with --total-lanes 1048576).

Workloads (config.workload in the JSON line)
  env_random  (default; BASELINE.json configs[1]) 65 536 lanes per GPU, built-in
              random-bitrate policy, 1 024 synthetic 1 000-pt traces.  A "step" is one
              chunk decision for every lane; min(--fuse, --steps) decisions share one kernel
              launch (config.fuse reports the number actually fused).
              metric = env-steps/s = lanes * K / time.  At N = 1 the same process then
              measures the other half of BASELINE.json's metric -- MPC combos/s at horizon 5
              (configs[2]) -- and reports it under "secondary" in the same JSON line.
  mpc         (configs[2], MPC half) 65 536 lanes x MPC horizon 5 over 6 rates:
              a step is one abr_mpc_select over all lanes; metric = combos/s.
  env_mpc     (configs[2]) MPC-driven rollout: every step = mpc_select + env step.

Timing: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both
sides, max over ranks.  When K < --min-timed-steps that region is repeated (`repeats`) and the
median repeat is reported, every repeat bracketed the same way.

Output (rank 0, stdout): the JSON line of the task's contract -- printed as soon as the headline is measured (at N = 1 together
with `roofline`, `cpu_baseline` and `selfcheck`, so that this first line is already complete), then printed AGAIN, grown by one
block, after each later block: `sustained`, `single_step`, `secondary`, `mpc_rollout`, `control` (N > 1), `strong_1048576`.  The
LAST line is the complete one; a run that is killed half-way has left every earlier line.  A block that raises is recorded as
{"error": ...} under its key and the run goes on (with more than one rank it then stops with exit code 3: the ranks must take
the same path, and a failed collective has broken the process group).  Every phase runs under a watchdog: when a collective or a launch does not
return within its bound (--phase-timeout; the process group itself is created with --dist-timeout) the rank says which phase
and which rank on stderr and exits with code 124 -- a fresh process is the only retry.

`selfcheck` (N = 1): sampled lanes of the LAST TIMED launch replayed through the C oracle (observation rows, rewards, done); a
mismatch fails the run.  `roofline`: dominant kernel, HIP-event timed inside the timed region; `binding` = what limits it, from
the committed SQ counters of the same kernel / lanes / fuse.  `cpu_baseline`: the C oracle on the host cores, bounded sample;
rank 0, N = 1 only; `affinity_cores` = cores visible, `cores` = threads used.

Environment overrides: everything named ABR_BENCH_* (and ABR_XCD_GROUPS, ABR_HIP_LIB) that is set is listed in config.overrides;
those that change the WORKLOAD (ABR_BENCH_MAX_BUFFER, _INTERVAL, _NTRACES, _STRONG_TOTAL, ABR_XCD_GROUPS) are refused unless
--allow-overrides is given, so a line without `overrides` is BASELINE's workload.
"""
import argparse
import json
import os
import sys
import threading
import time
from datetime import timedelta

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LADDER = [0.3, 0.75, 1.2, 1.85, 2.85, 4.3]
# (the overrides are diagnostics: max_buffer 1e9 = buffer_full never gates a download; a longer trace interval = fewer constant
# changes per download; fewer traces = a cache-residency diagnostic; a smaller strong-scaling job for the 2-rank rehearsal test)
V, L, START_UP = 48, 4.0, 8.0
MAX_BUFFER = float(os.environ.get("ABR_BENCH_MAX_BUFFER", "20.0"))
INTERVAL = float(os.environ.get("ABR_BENCH_INTERVAL", "1.0"))
WEIGHTS = [4.3, 1.0, 1.0, 0.1]
N_TRACES, TRACE_LEN = int(os.environ.get("ABR_BENCH_NTRACES", "1024")), 1000
STRONG_TOTAL = int(os.environ.get("ABR_BENCH_STRONG_TOTAL", "1048576"))   # configs[3] / north_star: the scaling curve's job size
WORKLOAD_OVERRIDES = ("ABR_BENCH_MAX_BUFFER", "ABR_BENCH_INTERVAL", "ABR_BENCH_NTRACES", "ABR_BENCH_STRONG_TOTAL", "ABR_XCD_GROUPS")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes/clk x 2 (FMA) x 2.4 GHz

# per-lane state bytes one launch reads and writes back (csrc/abr_env.hip: lane_load/lane_store
# + the scalars around them): 5 f64 (incl. the running episode's bitrate-variance sum, round 5) + 1 i64 + 13 i32 + 2 u8
STATE_BYTES = 5 * 8 + 8 + 13 * 4 + 2

KERNELS = {"async": "env_async_kernel<2>", "split": "env_split_kernel<2>", "split3": "env_split3_kernel<2>",
           "ring3": "env_ring3_kernel<2>", "pair3": "env_pair3_kernel<2>", "jump": "env_jump_kernel<2>",
           "tick": "env_advance_kernel<2>"}


# =====================================================================================================================
# harness: overrides, watchdog, progressive output, guarded blocks (no GPU needed: tests/test_bench_harness_cpu.py)
# =====================================================================================================================
def overrides_in_force(environ=None):
    """Every override of the environment that is set: {name: value}."""
    environ = os.environ if environ is None else environ
    return {k: environ[k] for k in sorted(environ)
            if (k.startswith("ABR_BENCH_") or k in ("ABR_XCD_GROUPS", "ABR_HIP_LIB")) and environ[k] != ""}


def check_overrides(allow, environ=None):
    """The overrides to record in config.overrides; raises SystemExit when one that changes the workload is set without
    --allow-overrides (a line printed under it would be indistinguishable from BASELINE's workload)."""
    ov = overrides_in_force(environ)
    bad = [k for k in ov if k in WORKLOAD_OVERRIDES]
    if bad and not allow:
        raise SystemExit(f"bench.py: {', '.join(bad)} change(s) the workload; pass --allow-overrides to run a diagnostic "
                         f"workload (the line then lists it under config.overrides)")
    return ov


class Watchdog:
    """A thread that ends the process when a phase outlives its bound: a collective whose peer died or stalled, a launch
    that never returns.  `with wd.phase("name", seconds): ...`; on expiry it prints which rank was in which phase for how
    long (stderr) and exits with code 124 through os._exit -- the main thread may be inside a blocking C call, and a fresh
    process is the only retry (no re-exec)."""

    EXIT_CODE = 124

    def __init__(self, rank=0, world=1, default_bound=150.0, poll=0.25, out=sys.stderr):
        self.rank, self.world, self.default_bound, self.poll, self.out = rank, world, float(default_bound), poll, out
        self._lock = threading.Lock()
        self._stack = []                 # (name, deadline, started)
        self.history = []                # phases that completed: (name, seconds)
        self._stop = False
        self._t = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self._t.start()

    class _Phase:
        def __init__(self, wd, name, bound):
            self.wd, self.name, self.bound = wd, name, bound

        def __enter__(self):
            now = time.monotonic()
            with self.wd._lock:
                self.wd._stack.append((self.name, now + self.bound, now))
            return self

        def __exit__(self, *exc):
            now = time.monotonic()
            with self.wd._lock:
                name, _, started = self.wd._stack.pop()
                self.wd.history.append((name, now - started))
            return False

    def phase(self, name, bound=None):
        return Watchdog._Phase(self, name, self.default_bound if bound is None else float(bound))

    def current(self):
        with self._lock:
            return self._stack[-1][0] if self._stack else None

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(self.poll)
            now = time.monotonic()
            with self._lock:
                late = [(n, now - s, d) for n, d, s in self._stack if now > d]
                stack = [n for n, _, _ in self._stack]
            if late:
                name, age, _ = late[0]
                msg = {"error": "timeout", "rank": self.rank, "world": self.world, "phase": name, "phase_stack": stack,
                       "seconds_in_phase": round(age, 1),
                       "completed_phases": [n for n, _ in self.history][-8:],
                       "note": "a collective or a launch did not return within its bound; every JSON line printed so far "
                               "stands; start a fresh process to retry"}
                try:
                    self.out.write("bench.py watchdog: " + json.dumps(msg) + "\n")
                    self.out.flush()
                    sys.stdout.flush()
                finally:
                    os._exit(self.EXIT_CODE)


class Emitter:
    """The growing JSON line: emit() prints it (rank 0) and flushes, so that every block measured so far is on stdout whatever
    happens next.  The last line printed is the complete one."""

    def __init__(self, rank=0, out=sys.stdout):
        self.rank, self.out, self.line, self.n_emitted = rank, out, {}, 0

    def update(self, **kw):
        self.line.update(kw)

    def emit(self):
        if self.rank == 0 and self.line:
            self.out.write(json.dumps(self.line) + "\n")
            self.out.flush()
            self.n_emitted += 1


def guarded(em, wd, key, fn, bound=None, keep_none=False, fatal=False):
    """Runs one optional block under the watchdog; its result goes under `key` of the line, an exception becomes
    {"error": ...} there (the headline and the blocks before it stand), and the grown line is printed.  fatal (more than one
    rank): the ranks of a run must take the same path, and a collective that failed has broken the process group -- so the
    error is recorded and printed as above, and then this rank exits with code 3 instead of going on to the next block."""
    failed = False
    try:
        with wd.phase(key, bound):
            res = fn()
    except Exception as e:                              # noqa: BLE001 -- a block must not take the line down with it
        import traceback
        failed = True
        res = {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc(limit=3).strip().splitlines()[-3:]}
    if res is not None or keep_none:
        em.update(**{key: res})
        em.emit()
    if failed and fatal:
        sys.stderr.write(f"bench.py: rank {wd.rank} of {wd.world}: block `{key}` failed ({res['error'][:300]}); with more than "
                         f"one rank the run cannot go on -- every JSON line printed so far stands\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(3)
    return res


# =====================================================================================================================
# workload pieces shared with tools/ and tests/
# =====================================================================================================================
def synth_traces(mixed=False):
    rng = np.random.default_rng(0)
    if mixed:
        lens = rng.integers(300, 3001, N_TRACES)
    else:
        lens = np.full(N_TRACES, TRACE_LEN)
    return [rng.uniform(0.2, 6.0, int(n)).astype(np.float32).astype(np.float64) for n in lens]


def lane_assignment(lane0, n, traces):
    # lane i -> trace i % n_traces.  ABR_XCD_GROUPS=8 selects the XCD-aware map
    # (workgroup w only reads traces t with t % 8 == w % 8); measured: no difference
    from abrsimulator_amd.sharding import lane_assignment as la
    return la(lane0, n, [len(t) for t in traces], xcd_groups=int(os.environ.get("ABR_XCD_GROUPS", "0")))


def host_cores():
    """(threads used, cores in the affinity mask).  The GPU box exposes every host core in the
    affinity mask but grants one GPU's share of CPU time (16 cores); ABR_BENCH_CORES overrides
    the thread count."""
    aff = len(os.sched_getaffinity(0))
    return max(1, min(aff, int(os.environ.get("ABR_BENCH_CORES", "16")))), aff


def cpu_baseline_env(traces, seed, budget_s=12.0):
    """The C oracle (oracle/abr_oracle.c) on every host core: same traces, same
    lane->trace map, same philox actions as the GPU run; a bounded lane sample."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    cores, aff = host_cores()
    cfg = O.env_cfg(LADDER, L, V, MAX_BUFFER, START_UP, INTERVAL, WEIGHTS, 1.0)

    def prep(lane0, n):
        tid, off = lane_assignment(lane0, n, traces)
        acts = np.stack([O.philox_action(seed, np.arange(lane0, lane0 + n), s, 0, len(LADDER))
                         for s in range(V)], 1).astype(np.int32)
        return tid, off, acts

    def run(inp):
        t0 = time.perf_counter()
        O.env_batch(cfg, traces, *inp)
        return time.perf_counter() - t0

    probe = run(prep(0, 256))                             # ~40 ms on one core
    per_core = max(256, int(256 * (budget_s / max(probe, 1e-4)) // 256 * 256))
    per_core = min(per_core, 16384)
    inputs = [prep(c * per_core, per_core) for c in range(cores)]   # untimed
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:                 # ctypes releases the GIL
        list(ex.map(run, inputs))
    wall = time.perf_counter() - t0
    n_steps = cores * per_core * V
    out = dict(value=n_steps / wall, unit="env-steps/s", cores=cores, affinity_cores=aff, kind="port",
               sample=f"{cores * per_core} lanes x {V}-chunk episodes ({n_steps} env-steps) of the "
                      f"same workload, C oracle -O2 -ffp-contract=off, one thread per core, "
                      f"{wall:.1f} s wall")
    # like-for-like interpreter figure (the reference is CPython): one core, a few lanes
    from oracle.pyloop import PyTickEnv, run_episode
    tid, off = lane_assignment(0, 16, traces)
    t0 = time.perf_counter()
    done = 0
    for i in range(16):
        e = PyTickEnv(LADDER, L, V, MAX_BUFFER, START_UP, INTERVAL, WEIGHTS, traces[tid[i]], int(off[i]))
        acts = [int(O.philox_action(seed, np.array([i]), s, 0, len(LADDER))[0]) for s in range(V)]
        run_episode(e, acts)
        done += V
        if time.perf_counter() - t0 > 8.0:
            break
    out["python_loop"] = dict(value=done / (time.perf_counter() - t0), unit="env-steps/s", cores=1,
                              sample=f"{done} env-steps, pure-CPython tick loop (oracle/pyloop.py)")
    return out


def selfcheck_env(sc, traces, lane0, seed):
    """Replays the sampled lanes of the LAST TIMED launch through the C oracle and compares every observation row,
    reward and done byte that launch wrote for them (float32 of the oracle's float64 value, `==`).  The launch covers
    decisions [n0, n0 + f) since the reset; all lanes run V-decision episodes in lock-step under auto_reset, so
    decision n is chunk n % V of episode n // V, whose actions are the counter-based policy's (the launch wrote no
    action buffer: the oracle.philox_action twin supplies them -- a wrong device action would show in every row)."""
    from oracle import oracle as O
    pick, n0, f = sc["pick"], sc["n0"], sc["f"]
    tid, off = lane_assignment(lane0, int(pick.max()) + 1, traces)
    tid, off = tid[pick], off[pick]
    cfg = O.env_cfg(LADDER, L, V, MAX_BUFFER, START_UP, INTERVAL, WEIGHTS, 1.0)
    P = len(pick)
    bad = total = 0
    first = None
    eps = {}
    for n in range(n0, n0 + f):
        e, c = divmod(n, V)
        if e not in eps:
            acts = np.stack([O.philox_action(seed, lane0 + pick, s_, e, len(LADDER)) for s_ in range(V)], 1).astype(np.int32)
            steps, bw, fin, _ = O.env_batch(cfg, traces, tid, off, acts)
            rew = O.step_rewards(steps["rebuffer_time"], steps["start_up_time"], fin["rebuffer_time"], fin["start_up_time"],
                                 acts, WEIGHTS, ladder=LADDER)
            eps[e] = (acts, steps, rew)
        acts, steps, rew = eps[e]
        r = n - n0
        if c < V - 1:      # the observation after decision c is the run() frame at call site c + 1
            q = c + 1
            want = [steps["chunk_id"][:, q], acts[:, c], steps["last_bandwidth"][:, q], steps["buffer_level"][:, q],
                    steps["global_time"][:, q], steps["play_time"][:, q], steps["rebuffer_time"][:, q], steps["start_up_time"][:, q]]
            dn = 0
        else:              # episode end under auto_reset: the fresh episode's first call site
            want = [steps["chunk_id"][:, 0], np.full(P, -1), np.zeros(P), steps["buffer_level"][:, 0],
                    steps["global_time"][:, 0], steps["play_time"][:, 0], steps["rebuffer_time"][:, 0], steps["start_up_time"][:, 0]]
            dn = 1
        for row in range(8):
            m = sc["obs"][r, row] != np.asarray(want[row], np.float64).astype(np.float32)
            bad += int(m.sum()); total += P
            if m.any() and first is None:
                first = f"decision {n} (episode {e}, chunk {c}) obs row {row} lane {int(lane0 + pick[np.argmax(m)])}"
        m = sc["reward"][r] != rew[:, c]
        bad += int(m.sum()); total += P
        if m.any() and first is None:
            first = f"decision {n} (episode {e}, chunk {c}) reward lane {int(lane0 + pick[np.argmax(m)])}"
        m = sc["done"][r] != dn
        bad += int(m.sum()); total += P
        if m.any() and first is None:
            first = f"decision {n} (episode {e}, chunk {c}) done lane {int(lane0 + pick[np.argmax(m)])}"
    return {"lanes": int(P), "decisions": int(f), "first_decision": int(n0), "elements": int(total), "mismatches": int(bad),
            "first_mismatch": first,
            "what": "obs rows (8), reward, done of the last timed launch for the sampled lanes == float32(C oracle), "
                    "incl. the first and last workgroup; actions from the oracle's philox twin"}


def cpu_baseline_mpc(budget_s=10.0):
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    cores, aff = host_cores()
    mc = O.mpc_cfg(len(LADDER), 5, V, L, MAX_BUFFER, 1.0, 4.3, 0.0)
    br = np.tile(np.array(LADDER), (V, 1))
    sz = br * L
    rng = np.random.default_rng(1)

    def run(n):
        t0 = time.perf_counter()
        O.mpc_select(mc, br, sz, rng.integers(0, V - 5, n), rng.integers(0, 6, n),
                     rng.uniform(0, MAX_BUFFER, n), np.full(n, 5.0), 5.0 / rng.uniform(0.5, 5, n))
        return time.perf_counter() - t0

    probe = run(64)
    per_core = int(min(65536, max(64, 64 * budget_s / max(probe, 1e-4))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda c: run(per_core), range(cores)))
    wall = time.perf_counter() - t0
    combos = cores * per_core * 6 ** 5
    return dict(value=combos / wall, unit="combos/s", cores=cores, affinity_cores=aff, kind="port",
                sample=f"{cores * per_core} lane decisions x 7776 combos, C oracle (literal "
                       f"objective per combo, no prefix sharing), {wall:.1f} s wall")


def _load_binding(kernel, lanes, fuse):
    """What actually binds the env kernel, from the committed SQ counter profile of the same
    (kernel, lanes, fuse): vector-issue utilisation and active lanes per vector instruction."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not name.endswith("_sq_counters.json"):
            continue
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if d.get("kernel") == kernel and d.get("lanes") == lanes and d.get("fuse") == fuse:
            b = dict(d.get("binding", {}))
            b["source"] = "profiles/" + name
            return b
    return None


def _load_traffic(workload, kernel, lanes, fuse):
    tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        t = json.load(open(tj)).get(workload)
    except Exception:
        return None
    for e in (t if isinstance(t, list) else [t]):      # one entry per profiled (kernel, lanes, fuse)
        if e and e.get("fuse") == fuse and e.get("lanes") == lanes and e.get("kernel") == kernel:
            return e
    return None


# =====================================================================================================================
# the run: one object holds what the blocks share; one method per block
# =====================================================================================================================
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4800)
    ap.add_argument("--warmup", type=int, default=480)
    ap.add_argument("--fuse", type=int, default=48,
                    help="chunk decisions per kernel launch (48 = one episode; 1 = one launch and one "
                         "all-gather per decision, BASELINE.json configs[3] taken literally)")
    ap.add_argument("--lanes-per-gpu", type=int, default=65536)
    ap.add_argument("--total-lanes", type=int, default=0,
                    help="strong scaling: split this many lanes over the GPUs.  The BASELINE.json scaling "
                         "curve is `--gpus N --total-lanes 1048576` for N = 1, 2, 4, 8")
    ap.add_argument("--workload", default="env_random", choices=["env_random", "mpc", "env_mpc"])
    ap.add_argument("--mixed-traces", action="store_true", help="trace lengths 300..3000 (configs[4])")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--impl", default="auto", choices=["auto", "async", "split", "split3", "ring3", "pair3", "jump", "tick"])
    ap.add_argument("--min-timed-steps", type=int, default=960,
                    help="when --steps is smaller than this the timed region of exactly --steps steps is "
                         "repeated (each repeat bracketed by barrier + synchronize) and the MEDIAN repeat "
                         "is reported, so that one sub-millisecond launch is not the whole sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="env_random, N=1: skip the sustained block")
    ap.add_argument("--no-single-step", action="store_true", help="env_random, N=1: skip the single_step block (abr_env_step launches)")
    ap.add_argument("--sustained-seconds", type=float, default=1.0)
    ap.add_argument("--no-mpc-rollout", action="store_true",
                    help="env_random, N=1: skip the mpc_rollout block (BASELINE.json configs[2] composed and configs[4]'s per-rank shape)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="env_random, N=1: skip the MPC combos/s half of BASELINE.json's metric")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of (obs, reward)")
    ap.add_argument("--no-split-launch", action="store_true",
                    help="N>1: keep a timed region of --steps <= --fuse decisions as ONE launch (default: two launches "
                         "of half the decisions, so that the first all-gather overlaps the second launch)")
    ap.add_argument("--no-strong", action="store_true",
                    help="env_random: skip the strong_1048576 block (BASELINE.json configs[3]: 1 048 576 lanes "
                         "split over the ranks of this run)")
    ap.add_argument("--graph", action="store_true",
                    help="env_random, N=1: capture one launch in a HIP graph and replay it "
                         "(removes the host launch path; matters at --fuse 1)")
    ap.add_argument("--allow-overrides", action="store_true",
                    help="accept the ABR_BENCH_* / ABR_XCD_GROUPS overrides that change the workload (diagnostics); they are "
                         "listed under config.overrides either way")
    ap.add_argument("--dist-timeout", type=float, default=90.0,
                    help="N>1: seconds the process group waits for a peer (rendezvous and every collective)")
    ap.add_argument("--phase-timeout", type=float, default=150.0,
                    help="seconds any single phase of the run (a timed region, a block) may take before the watchdog ends the "
                         "process with exit code 124, saying which rank was in which phase")
    return ap.parse_args(argv)


class Run:
    def __init__(self, a):
        self.a = a
        # ---- launcher environment first: nothing below may touch the GPU before this is settled ----
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if a.gpus != self.world and self.world > 1:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={self.world}")
        if a.gpus > 1 and self.world == 1:
            raise SystemExit("launch N>1 through torch.distributed.run (one rank per GPU)")
        if a.total_lanes and a.total_lanes % self.world:
            raise SystemExit("--total-lanes must divide evenly over the GPUs (equal gather shapes)")
        self.overrides = check_overrides(a.allow_overrides)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: read when HSA initialises
        # kernel arguments in device memory instead of host-coherent memory (read when HIP initialises): the role-split
        # kernels re-read their parameter block inside the iteration loop; same box, interleaved: +2.5 % at 48 decisions
        # per launch, +4.5 % at 20 (profiles/r06_ab_dev_kernarg.txt).  The package sets the same default on import.
        os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
        # rehearsal knobs (a 1-GPU box cannot run RCCL with 2 ranks): ABR_BENCH_ONE_DEVICE=1 puts every
        # rank on cuda:0, ABR_BENCH_BACKEND=gloo swaps the backend.  The driver's runs use neither.
        if os.environ.get("ABR_BENCH_ONE_DEVICE") == "1":
            self.local_rank = 0
        self.wd = Watchdog(self.rank, self.world, a.phase_timeout)
        self.em = Emitter(self.rank)

    # ---- process group, device, environment ----
    def setup(self):
        a = self.a
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.backend = "none"
        # ABR_BENCH_FORCE_DIST=1: a ONE-rank process group, so that the RCCL code path (communicator,
        # all_gather_into_tensor on the side stream, events) runs on a one-GPU box too
        self.force_dist = self.world == 1 and os.environ.get("ABR_BENCH_FORCE_DIST") == "1"
        if self.force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        self.distributed = self.world > 1 or self.force_dist
        if self.distributed:
            self.backend = os.environ.get("ABR_BENCH_BACKEND", "nccl")
            to = timedelta(seconds=a.dist_timeout)
            with self.wd.phase("init_process_group", a.dist_timeout + 30):
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=self.dev, timeout=to)      # "nccl" IS RCCL on ROCm
                else:
                    dist.init_process_group(self.backend, timeout=to)

        import abrsimulator_amd as A
        from abrsimulator_amd._lib import OBS_DIM
        from abrsimulator_amd.sharding import shard_range
        self.A, self.OBS_DIM, self.shard_range = A, OBS_DIM, shard_range
        if a.total_lanes:
            self.lane0, self.N = shard_range(a.total_lanes, self.world, self.rank)       # strong scaling
        else:
            self.N = a.lanes_per_gpu                                                     # weak scaling (default)
            self.lane0 = self.rank * self.N
        self.traces = synth_traces(a.mixed_traces)
        tid, off = lane_assignment(self.lane0, self.N, self.traces)
        self.mpd = A.MPD(V, L, MAX_BUFFER, START_UP, A.Chunk(LADDER))
        self.env = A.BatchedABREnv(self.mpd, A.QOEMetric(*WEIGHTS), A.NetworkInfo(INTERVAL, self.traces), self.N,
                                   device=self.dev, auto_reset=True, lane_id_base=self.lane0, impl=a.impl)
        self.env.reset(torch.from_numpy(tid), torch.from_numpy(off))
        self.K, self.W = a.steps, a.warmup
        self.repeats = max(1, -(-a.min_timed_steps // self.K)) if self.K < a.min_timed_steps else 1

    def barrier(self):
        """dist.barrier (bounded by the process group's timeout and by the watchdog phase around the caller) + synchronize."""
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def timed_region(self, runner, n_steps, events=True):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; max over ranks.  events: every launch of the
        region between two HIP events (the kernel's own duration; costs a single-launch region ~7 us); False: launches only."""
        torch, dist = self.torch, self.dist
        self.barrier()
        t0 = time.perf_counter()
        runner(n_steps, "events" if events else "plain")
        self.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.dev)
        if self.distributed:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def measure(self, runner, n_steps, reps, phase=None):
        """`reps` x (one region with HIP events around every launch, one region with the launches only), interleaved, each
        bracketed as timed_region() says.  Returns (times of the plain regions, times of the instrumented ones): the plain
        ones are what a value is computed from, the instrumented ones fill the runner's event list (launch_stats)."""
        plain, inst = [], []
        for r in range(reps):
            if phase:
                with self.wd.phase(f"{phase}_with_events[{r}]"):
                    inst.append(self.timed_region(runner, n_steps, events=True))
                with self.wd.phase(f"{phase}[{r}]"):
                    plain.append(self.timed_region(runner, n_steps, events=False))
            else:
                inst.append(self.timed_region(runner, n_steps, events=True))
                plain.append(self.timed_region(runner, n_steps, events=False))
        return plain, inst

    @staticmethod
    def launch_stats(events):
        ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
        return float(np.mean(ms)) * 1e-3, float(np.mean([f for _, _, f in events]))

    # ---- runners ----
    def mpc_setup(self):
        A, a = self.A, self.a
        player = A.EnvPlayer(self.env, mpd=A.MPD(V, L, MAX_BUFFER, START_UP, [A.Chunk(LADDER, [b * L for b in LADDER])] * V),
                             qoe=A.QOEMetric(4.3, 1.0, 0.0))
        ctl = A.BatchedMPCController(player, horizon=5, clip_horizon=True, device=self.dev)
        # give every lane a history first (the reference divides by zero on an empty one, D13)
        self.env.step_random(3, a.seed)
        return player, ctl

    def mpc_runner(self, ctl, events, drive_env):
        torch, N, dev = self.torch, self.N, self.dev
        mpc_out = dict(obs=torch.empty(1, self.OBS_DIM, N, dtype=torch.float32, device=dev),
                       reward=torch.empty(1, N, dtype=torch.float32, device=dev),
                       done=torch.empty(1, N, dtype=torch.uint8, device=dev), actions=None)

        pool = []

        def run(n_steps, timed):
            if not timed:                                 # warm-up call: HIP events are created on their first record()
                while len(pool) < 2 * n_steps + 64:
                    e = torch.cuda.Event(enable_timing=True); e.record(); pool.append(e)
            for _ in range(n_steps):
                # (no history restore between selects: every call grows the lanes' throughput history by
                # `horizon` predictions, D9, exactly as repeated next_bitrate() calls do in the reference)
                if timed == "events":
                    e0 = pool.pop() if pool else torch.cuda.Event(enable_timing=True); e0.record()
                if drive_env:
                    self.env.step_mpc(ctl, 1, out=mpc_out)       # K3 + K1 on the device, no host glue
                else:
                    ctl.next_bitrate()
                if timed == "events":
                    e1 = pool.pop() if pool else torch.cuda.Event(enable_timing=True); e1.record()
                    events.append((e0, e1, 1))
        return run

    def make_random_runner(self, env_, N_, F_, events_, use_graph, gather_on=True, total_lanes=None):
        """The env_random step loop over `env_`, through the package's ShardedABREnv (abrsimulator_amd/sharding.py): launches
        of F_ fused decisions into double-buffered slabs; with more than one rank THE one collective of the path per launch
        -- ONE all-gather of the packed slab [final observation (8 x N) | rewards (F x N)] on a side stream, overlapped with
        the next launch.  The intermediate observations of a fused launch are consumed on-device by the built-in policy and
        stay in the local slab; at --fuse 1 every observation is gathered (configs[3] literally).
        total_lanes: the job's size under STRONG scaling, so that uneven shards agree on the largest one (the staging
        buffer and the all-gather's size); None = weak scaling, every rank holds N_ lanes."""
        from abrsimulator_amd.sharding import ShardedABREnv
        torch, a, dev = self.torch, self.a, self.dev
        gather = self.distributed and not a.no_gather and gather_on
        kw = dict(total_lanes=total_lanes) if total_lanes else dict(lanes_per_rank=N_)
        sh = ShardedABREnv(None, None, env_.network_info, fuse=F_, device=dev, gather=gather, env=env_,
                           rank=self.rank, world=self.world, **kw)
        assert sh.n_lanes == N_, (sh.n_lanes, N_)
        gat_ = sh._gather
        graph = None
        if use_graph:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                env_.step_random(F_, a.seed, out=sh._outs[0])   # warm-up on a side stream before capture
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                env_.step_random(F_, a.seed, out=sh._outs[0])

        pool = []      # HIP events are created on their first record(): do that outside the timed region
        info = dict(n_done=0, last=None, bufs=sh._outs, sharded=sh)     # decisions run so far; (buffer, decisions, first decision) of the last launch
        K = self.K

        def take():
            return pool.pop() if pool else torch.cuda.Event(enable_timing=True)

        def run_(n_steps, timed):
            """timed: False = warm-up; "events" = a region whose every launch sits between two HIP events (the kernel's
            duration); True / "plain" = a region with nothing but the launches in it (the headline)."""
            left = n_steps
            ev_on = timed == "events"
            if not timed:                                 # warm-up call: stock the pool for the timed calls
                need = 2 * (max(1, -(-a.min_timed_steps // max(K, 1))) + 1) * (K // F_ + 2)
                while len(pool) < need:
                    e = torch.cuda.Event(enable_timing=True); e.record(); pool.append(e)
            while graph is not None and left >= F_:
                if ev_on:
                    e0 = take(); e0.record()
                graph.replay()
                if ev_on:
                    e1 = take(); e1.record()
                    events_.append((e0, e1, F_))
                info["last"] = (0, F_, info["n_done"]); info["n_done"] += F_
                left -= F_
            while left > 0:
                f = min(F_, left)
                evp = (take(), take()) if ev_on else None
                b = sh._it & 1
                sh.step_random(f, a.seed, events=evp)     # wait for the slab, launch, enqueue the all-gather
                if ev_on:
                    events_.append((evp[0], evp[1], f))
                info["last"] = (b, f, info["n_done"]) if f == F_ else None
                info["n_done"] += f
                left -= f
            sh.finish()
        run_.info = info
        return run_, gat_

    # ---- block: the headline ----
    def headline(self):
        a, K, W = self.a, self.K, self.W
        self.ev = []
        self.gat = None
        self.F1 = self.F = 1
        if a.workload == "env_random":
            self.F1 = max(1, min(a.fuse, K))           # decisions fused into one launch at N = 1 (what BENCH runs)
            self.F = self.F1
            if (self.distributed and not a.no_gather and not a.no_split_launch and K <= a.fuse and K >= 2 and K % 2 == 0):
                # more than one rank and the whole timed region would be ONE launch: its all-gather would have
                # nothing to hide behind (the bracket closes right after it).  Two launches of K/2 decisions put
                # the first gather under the second launch; only the second, smaller one stays exposed.  (Even K
                # only: both launches then use the pre-bound slabs and both are gathered.)  The `control` block of
                # the JSON line separates what this change of launch shape costs from what the collective costs.
                self.F = K // 2
            self.impl = self.env.effective_impl(fused=self.F > 1)     # launches of ONE decision resolve differently (abr_env.h)
            self.env_kernel = KERNELS[self.impl]
            self.run, self.gat = self.make_random_runner(self.env, self.N, self.F, self.ev, a.graph and self.world == 1,
                                                         total_lanes=a.total_lanes or None)
            self.units_per_step = self.N * self.world if not a.total_lanes else a.total_lanes
            self.unit, self.metric = "env-steps/s", "env_steps_per_sec"
        else:
            # env_mpc launches K1 one decision at a time; `mpc` launches no env kernel
            self.impl = self.env.effective_impl(fused=False) if a.workload == "env_mpc" else None
            self.env_kernel = None
            _, ctl = self.mpc_setup()
            self.run = self.mpc_runner(ctl, self.ev, a.workload == "env_mpc")
            if a.workload == "mpc":
                self.units_per_step = self.N * self.world * 6 ** 5
                self.unit, self.metric = "combos/s", "mpc_combos_per_sec"
            else:
                self.units_per_step = self.N * self.world
                self.unit, self.metric = "env-steps/s", "env_steps_per_sec_mpc_policy"
        with self.wd.phase("warmup"):
            self.run(W, False)
        # The headline's regions hold the K steps and NOTHING else.  The kernel's duration (roofline) is measured live, with HIP
        # events around every launch, in regions of the same K steps and the same bracket INTERLEAVED with them: the two event
        # records cost a region that is one launch ~7 us (5 % of the driver's 20-decision region, tools/gpu_event_cost.py), and
        # that is instrumentation, not workload.  Both series are in the line (repeat_seconds, repeat_seconds_with_events).
        self.times, self.times_ev = self.measure(self.run, K, self.repeats, phase="timed_region")
        self.elapsed = float(np.median(self.times))
        self.avg_launch_s, self.f_per_launch = self.launch_stats(self.ev)
        roof = self.env_roofline() if a.workload == "env_random" else self.mpc_roofline(self.avg_launch_s)
        gathering = self.distributed and not a.no_gather and a.workload == "env_random"
        F, N = self.F, self.N
        config = {"workload": a.workload, "lanes_per_gpu": N, "total_lanes": self.units_per_step if a.workload != "mpc" else N * self.world,
                  "fuse": F, "launches_per_region": -(-K // F), "fuse_at_n1": self.F1 if a.workload == "env_random" else 1,
                  "impl": self.impl,
                  "history_restore": False if a.workload != "env_random" else None,
                  "video_length": V, "chunk_length_s": L, "n_rates": len(LADDER),
                  "traces": f"{N_TRACES} x " + ("300..3000" if a.mixed_traces else str(TRACE_LEN)),
                  "policy": "random(philox)" if a.workload == "env_random" else "mpc_h5",
                  "hip_graph": bool(a.graph and self.world == 1 and a.workload == "env_random"),
                  "auto_reset": True,
                  "collective": (f"1 all_gather_into_tensor per launch of the packed slab [final obs 8xN | "
                                 f"rewards {F}xN] float32 = {(8 + F) * N * 4} B per rank, overlapped "
                                 f"with the next launch; backend {self.backend}; issued "
                                 f"{self.gat.n_collectives if self.gat else 0}x" if gathering else "none")}
        if self.overrides:
            config["overrides"] = self.overrides
        self.em.update(metric=self.metric, value=self.units_per_step * K / self.elapsed, unit=self.unit, n_gpus=self.world,
                       steps=K, warmup=W, ms_per_step=self.elapsed / K * 1e3, higher_is_better=True,
                       scaling="strong" if a.total_lanes else "weak", vs_baseline=None, dtype="f64", data="synthetic",
                       repeats=self.repeats, repeat_seconds=self.times, repeat_seconds_with_events=self.times_ev,
                       value_with_events=self.units_per_step * K / float(np.median(self.times_ev)),
                       config=config, roofline=roof, cpu_baseline=None)

    def env_roofline(self):
        # algorithmic bytes per launch (DESIGN.md "Roofline"): per lane, state in + out once per
        # launch; per decision: obs 32 + reward 4 + done 1 out, previous_bitrates 1 +
        # previous_bandwidths 8 appended, and the trace points walked (8 B bandwidth + 4 B
        # interval-end tick each; 407 ticks/decision at interval 1 s -> 4.07 points).
        N, f_per_launch, avg_launch_s = self.N, self.f_per_launch, self.avg_launch_s
        pts = 4.07 * (8 + 4)
        per_decision = 32 + 4 + 1 + 1 + 8 + pts
        alg_bytes = N * (2 * STATE_BYTES + f_per_launch * per_decision)
        roof = dict(bound="hbm", kernel=self.env_kernel,
                    achieved=alg_bytes / avg_launch_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                    traffic=None, avg_launch_us=avg_launch_s * 1e6,
                    algorithmic_bytes_per_launch=alg_bytes, decisions_per_launch=f_per_launch,
                    note="tick-exact semantics make this kernel vector-issue / latency-bound, not "
                         "HBM-bound (SURVEY.md 8d); the HBM fraction is reported as mandated, and "
                         "`binding` names the resource that actually limits it")
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["frac_formula"] = ("fused-launch bytes: per decision 32 obs + 4 reward + 1 done + 9 history + 4.07 trace points x 12 "
                                "= 94.8 B, plus 2 x 102 B of lane state ONCE per launch")
        # SURVEY.md 8(d) literally: state read + write (2 x 96) + action in 4 + obs/reward/done out 29 + 4 B per trace point
        # walked, charged PER STEP -- what a one-decision-per-launch kernel moves; a fused launch does not re-read the state
        survey_bytes = N * f_per_launch * (192 + 4 + 29 + 4 * 4.07)
        roof["frac_survey_8d"] = survey_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS
        roof["frac_survey_8d_formula"] = "SURVEY.md 8(d): (192 + 4 + 29 + 4 x points walked) B per step = 241.3 B, state traffic charged per step"
        t = _load_traffic("env_random", self.env_kernel, N, int(round(f_per_launch)))
        if t:
            roof["traffic"] = t["bytes_per_launch"]
            roof["traffic_source"] = t.get("source")
        b = _load_binding(self.env_kernel, N, int(round(f_per_launch)))
        if b:
            roof["binding"] = b
        return roof

    def mpc_roofline(self, launch_s):
        # K3 is fp64-VALU-bound.  Executed work with prefix sharing: 7 776 leaves x
        # 9 flop + 1 554 inner nodes x 13 flop = 90.2 kflop per lane decision (82.4 with the x1.0
        # weight dropped; the reference's
        # from-scratch formulation is 99 flop x 7 776 combos = 770 kflop).  No FMA may
        # be used (-ffp-contract=off is the parity contract), so 50 % of the FMA peak is the ceiling.
        # (variance_weight is exactly 1.0 here, so the kernel drops that multiplication: 8 per leaf)
        N = self.N
        flops = N * (7776 * 8.0 - 1296 * 2.0 + 1554 * 13.0)   # 1 296 leaves repeat their parent digit: variance term is +0.0, skipped
        roof = dict(bound="valu_fp64", kernel="mpc_select_kernel<5,6>",
                    achieved=flops / launch_s / 1e12, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s",
                    traffic=None, avg_launch_us=launch_s * 1e6,
                    reference_formulation_tflops=N * 6 ** 5 * 99.0 / launch_s / 1e12,
                    note="executed fp64 flop (prefix-sharing DFS) against the FMA peak; "
                         "add/mul only, so 0.5 is the ceiling")
        roof["frac"] = roof["achieved"] / roof["peak"]
        t = _load_traffic("mpc", "mpc_select_kernel<5,6>", N, 1)
        if t:
            roof["traffic"] = t["bytes_per_launch"]
            roof["traffic_source"] = t.get("source")
        return roof

    # ---- block: self-check (N = 1).  Part 1 keeps what the LAST TIMED launch wrote for a sample of lanes -- its observation
    #      rows, rewards and done bytes -- before anything overwrites the slab; part 2 replays those lanes through the oracle ----
    def selfcheck_capture(self):
        torch, N = self.torch, self.N
        info = getattr(self.run, "info", None)
        if not (info and info["last"]):
            return None
        b_, f_, n0_ = info["last"]
        rng_ = np.random.default_rng(5)
        pick = np.unique(np.concatenate([np.arange(8), np.arange(N - 8, N), [63, 64, 127, 128],
                                         rng_.integers(0, N, 300)])).astype(np.int64)
        pick = pick[(pick >= 0) & (pick < N)]
        pk = torch.from_numpy(pick).to(self.dev)
        bo = info["bufs"][b_]
        return dict(pick=pick, n0=n0_, f=f_, obs=bo["obs"][:f_].index_select(2, pk).cpu().numpy(),
                    reward=bo["reward"][:f_].index_select(1, pk).cpu().numpy(),
                    done=bo["done"][:f_].index_select(1, pk).cpu().numpy())

    # ---- block: sustained (N = 1): at least a second of back-to-back launches of the SAME shape in ONE bracket, so that
    #      clocks and thermals have settled and an outside sampler sees the GPU busy; the headline stays what it was ----
    def sustained(self):
        torch, a, N, F = self.torch, self.a, self.N, self.F
        info = self.run.info
        n_l = int(np.ceil(a.sustained_seconds * 1.1 / max(self.avg_launch_s, 1e-6)))
        per_mark = max(1, int(0.1 / max(self.avg_launch_s, 1e-6)))          # an event every ~100 ms
        bufs_ = info["bufs"]
        marks = []
        self.barrier()
        t0 = time.perf_counter()
        for j in range(n_l):
            if j % per_mark == 0:
                e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((j, e))
            self.env.step_random(F, a.seed, out=bufs_[j & 1])
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((n_l, e))
        self.barrier()
        el_s = time.perf_counter() - t0
        info["n_done"] += n_l * F; info["last"] = None
        win = [(j1 - j0) * F * N / (e0.elapsed_time(e1) * 1e-3) for (j0, e0), (j1, e1) in zip(marks[:-1], marks[1:])
               if j1 - j0 == per_mark]
        return {"value": N * F * n_l / el_s, "unit": self.unit, "seconds": el_s, "launches": n_l, "fuse": F,
                "ms_per_step": el_s / (n_l * F) * 1e3,
                "first_100ms_value": win[0] if win else None, "last_100ms_value": win[-1] if win else None,
                "last_over_first": (win[-1] / win[0]) if len(win) >= 2 else None,
                "windows_100ms": len(win), "min_window_value": min(win) if win else None,
                "note": "one bracket (barrier + synchronize on both sides) around all launches; the windows are HIP "
                        "events recorded every ~100 ms inside it"}

    # ---- block: the reset()/get_video_chunk(quality) surface itself (N = 1): ONE decision per launch with the caller's
    #      actions, abr_env_step (Simulator.py:155's call site turned inside out) ----
    def single_step(self):
        torch, N = self.torch, self.N
        acts1 = torch.randint(0, len(LADDER), (N,), dtype=torch.int32, device=self.dev)
        for _ in range(50):
            self.env.step(acts1)
        n1 = 400
        self.barrier(); t0 = time.perf_counter()
        for _ in range(n1):
            self.env.step(acts1)
        self.barrier()
        el1 = time.perf_counter() - t0
        return {"metric": "env_steps_per_sec_single_decision_launches", "value": N * n1 / el1, "unit": "env-steps/s",
                "launches": n1, "us_per_launch": el1 / n1 * 1e6, "impl": self.env.effective_impl(fused=False),
                "call": "abr_env_step(actions): one decision per launch, caller-supplied actions (a fixed random vector)"}

    # ---- block (N > 1 only): what the collective costs and what the launch shape costs, measured in the same process
    #      with the same bracket, so that the scaling curve can be read: (1) the SAME launches without the
    #      all-gather; (2) the N = 1 launch shape (one launch of F1 decisions) without the all-gather ----
    def control_block(self, env_, N_, F_used, total_lanes_, total_arg=None):
        out = {}
        for name, f_ in (("same_launches_no_gather", F_used), ("n1_launch_shape_no_gather", self.F1)):
            evc = []
            runc, _ = self.make_random_runner(env_, N_, f_, evc, False, gather_on=False, total_lanes=total_arg)
            runc(min(self.W, 2 * f_), False)
            reps = max(1, min(self.repeats, 12))
            tc = float(np.median(self.measure(runc, self.K, reps)[0]))
            lsc, _ = self.launch_stats(evc)
            out[name] = {"value": total_lanes_ * self.K / tc, "ms_per_step": tc / self.K * 1e3, "fuse": f_,
                         "launches_per_region": -(-self.K // f_), "avg_launch_us": lsc * 1e6, "repeats": reps}
        return out

    # ---- block: the other half of BASELINE.json's metric, same process, same JSON line ----
    def secondary(self):
        N = self.N
        ev2 = []
        _, ctl = self.mpc_setup()
        run2 = self.mpc_runner(ctl, ev2, False)
        K2, W2 = 100, 100
        run2(W2, False)
        # five regions of K2 selects, the median reported and all five kept: the first region after the env
        # workload runs on a clock that is still settling (a 100-select region is 20 ms)
        times2, _ = self.measure(run2, K2, 5)
        el2 = float(np.median(times2))
        ls2, _ = self.launch_stats(ev2)
        return {"metric": "mpc_combos_per_sec", "value": N * 6 ** 5 * K2 / el2, "unit": "combos/s",
                "steps": K2, "warmup": W2, "ms_per_step": el2 / K2 * 1e3,
                "repeats": len(times2), "repeat_seconds": times2,
                "config": {"workload": "mpc", "lanes_per_gpu": N, "n_rates": 6, "horizon": 5,
                           "combos_per_lane": 6 ** 5, "predictor": "harmonic",
                           # every select grows the lanes' history by `horizon` predictions (D9), as repeated
                           # next_bitrate() calls do in the reference; round 2's loop restored it between selects
                           "history_restore": False},
                "roofline": self.mpc_roofline(ls2)}

    # ---- block: BASELINE.json configs[3] / north_star "1/2/4/8-GPU scaling curve on 1 048 576 lanes": the SAME job
    #      split over however many ranks this run has, measured in this process after the headline metric
    #      (which stays 65 536 lanes per GPU so that N = 1 is BENCH's number).  The driver's back-to-back
    #      N = 1, 2, 4, 8 runs therefore yield both curves: `value` (weak) and `strong_1048576.value`. ----
    def strong(self):
        torch, A, a, F, K = self.torch, self.A, self.a, self.F, self.K
        lane0s, Ns = self.shard_range(STRONG_TOTAL, self.world, self.rank)
        tids, offs = lane_assignment(lane0s, Ns, self.traces)
        env_s = A.BatchedABREnv(self.mpd, A.QOEMetric(*WEIGHTS), A.NetworkInfo(INTERVAL, self.traces), Ns, device=self.dev,
                                auto_reset=True, lane_id_base=lane0s, impl=a.impl)
        env_s.reset(torch.from_numpy(tids), torch.from_numpy(offs))
        ev_s = []
        run_s, gat_s = self.make_random_runner(env_s, Ns, F, ev_s, False, total_lanes=STRONG_TOTAL)
        run_s(min(self.W, 2 * F), False)
        reps_s = max(1, min(self.repeats, 12))
        times_s, _ = self.measure(run_s, K, reps_s)
        el_s = float(np.median(times_s))
        ls_s, _ = self.launch_stats(ev_s)
        control_s = None
        if self.distributed and not a.no_gather:
            control_s = self.control_block(env_s, Ns, F, STRONG_TOTAL, total_arg=STRONG_TOTAL)
        n_max = max(self.shard_range(STRONG_TOTAL, self.world, r)[1] for r in range(self.world))
        out = {"metric": "env_steps_per_sec", "value": STRONG_TOTAL * K / el_s, "unit": "env-steps/s",
               "scaling": "strong", "n_gpus": self.world, "total_lanes": STRONG_TOTAL, "lanes_per_gpu": Ns,
               "steps": K, "ms_per_step": el_s / K * 1e3, "repeats": reps_s, "fuse": F,
               "launches_per_region": -(-K // F), "control": control_s,
               "impl": env_s.effective_impl(fused=F > 1), "avg_launch_us": ls_s * 1e6,
               "collective": (f"1 all_gather_into_tensor per launch, {(8 + F) * n_max * 4} B per rank; issued "
                              f"{gat_s.n_collectives}x" if gat_s else "none")}
        del env_s
        return out

    # ---- block: the MPC-driven rollout (N = 1): BASELINE.json configs[2] composed (65 536 envs x MPC horizon 5,
    #      abr_env_step_mpc) and configs[4]'s per-rank shape (131 072 lanes of the 1 048 576-lane job, rank 7's lane ids, mixed
    #      300-3 000-point traces).  One timed region = ONE 48-decision episode of every lane in one abr_env_step_mpc call; the
    #      K3 / K1 split comes from HIP events around the two halves of a host-loop pass over the same decisions. ----
    def mpc_rollout(self):
        torch, A, a, dev, OBS_DIM = self.torch, self.A, self.a, self.dev, self.OBS_DIM
        out = {}
        for name, lanes_r, mixed_r, base_r in (("configs2_65536", 65536, False, 0),
                                               ("configs4_rank7_131072_mixed", 131072, True, 7 * 131072)):
            try:
                out[name] = self._mpc_rollout_one(lanes_r, mixed_r, base_r)
            except Exception as e:                          # noqa: BLE001 -- the other shape still counts
                out[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        return out

    def _mpc_rollout_one(self, lanes_r, mixed_r, base_r):
        torch, A, a, dev, OBS_DIM = self.torch, self.A, self.a, self.dev, self.OBS_DIM
        tr_r = synth_traces(mixed_r)
        tid_r, off_r = lane_assignment(base_r, lanes_r, tr_r)
        env_r = A.BatchedABREnv(self.mpd, A.QOEMetric(*WEIGHTS), A.NetworkInfo(INTERVAL, tr_r), lanes_r, device=dev,
                                auto_reset=True, lane_id_base=base_r, impl=a.impl)
        env_r.reset(torch.from_numpy(tid_r), torch.from_numpy(off_r))
        ctl_r = A.BatchedMPCController(
            A.EnvPlayer(env_r, mpd=A.MPD(V, L, MAX_BUFFER, START_UP, [A.Chunk(LADDER, [b * L for b in LADDER])] * V),
                        qoe=A.QOEMetric(4.3, 1.0, 0.0)), horizon=5, clip_horizon=True, device=dev)
        out_r = dict(obs=torch.empty(V, OBS_DIM, lanes_r, dtype=torch.float32, device=dev),
                     reward=torch.empty(V, lanes_r, dtype=torch.float32, device=dev),
                     done=torch.empty(V, lanes_r, dtype=torch.uint8, device=dev),
                     actions=torch.empty(V, lanes_r, dtype=torch.int32, device=dev))

        def region(n_):
            env_r.step_mpc(ctl_r, n_, out=out_r)
        region(V)                                            # one episode of warm-up
        times_r = []
        for _ in range(5):
            self.barrier(); t0 = time.perf_counter(); region(V); self.barrier()
            times_r.append(time.perf_counter() - t0)
        el_r = float(np.median(times_r))
        ends = int(out_r["done"][-1].sum().item())           # every lane ends its episode at the last decision
        # the split: the same decisions as select + step from the host, HIP events around each half
        evs = []
        for _ in range(V):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e2 = torch.cuda.Event(enable_timing=True)
            e0.record(); act = ctl_r.next_bitrate(); e1.record()
            env_r.step(torch.clamp(act, min=0))      # D13 at chunk 0: "no decision" -> rate 0, as the fused call does
            e2.record()
            evs.append((e0, e1, e2))
        torch.cuda.synchronize(dev)
        k3 = float(np.median([x.elapsed_time(y) for x, y, _ in evs])) * 1e3
        k1 = float(np.median([y.elapsed_time(z) for _, y, z in evs])) * 1e3
        return {
            "metric": "env_steps_per_sec_mpc_policy", "value": lanes_r * V / el_r, "unit": "env-steps/s",
            "combos_per_sec": lanes_r * V * 6 ** 5 / el_r, "lanes": lanes_r, "lane_id_base": base_r,
            "decisions_per_region": V, "us_per_decision": el_r / V * 1e6, "repeats": len(times_r), "repeat_seconds": times_r,
            "traces": f"{N_TRACES} x " + ("300..3000" if mixed_r else str(TRACE_LEN)),
            "episodes_ended_at_last_decision": ends, "call": "abr_env_step_mpc(n_steps=48): K3 + K1 per decision, no host work between",
            "split_host_loop": {"k3_select_us": k3, "k1_step_us": k1, "k1_impl": env_r.effective_impl(fused=False),
                                "note": "medians of HIP events around next_bitrate() and step() (+ one clamp kernel) of a host "
                                        "loop over one more episode"}}

    # ---- the whole run ----
    def main(self):
        a, em, wd = self.a, self.em, self.wd
        with wd.phase("setup", max(a.phase_timeout, 300.0)):     # the first import of torch on a fresh box pages in for minutes
            self.setup()
        self.headline()
        env_rand, n1 = a.workload == "env_random", self.world == 1
        # N = 1: the self-check and the CPU baseline belong to the FIRST line printed, so that it is complete by itself
        selfcheck = None
        if self.rank == 0 and n1 and not a.no_cpu_baseline:
            sc_in = None
            if env_rand:
                try:
                    with wd.phase("selfcheck_capture"):
                        sc_in = self.selfcheck_capture()
                except Exception as e:                      # noqa: BLE001
                    em.update(selfcheck={"error": f"{type(e).__name__}: {e}"})
            if sc_in is not None:
                try:
                    with wd.phase("selfcheck", 300.0):
                        selfcheck = selfcheck_env(sc_in, self.traces, self.lane0, a.seed)
                    em.update(selfcheck=selfcheck)
                except Exception as e:                      # noqa: BLE001
                    em.update(selfcheck={"error": f"{type(e).__name__}: {e}"})
            try:
                with wd.phase("cpu_baseline", 300.0):
                    em.update(cpu_baseline=cpu_baseline_mpc() if a.workload == "mpc" else cpu_baseline_env(self.traces, a.seed))
            except Exception as e:                          # noqa: BLE001
                em.update(cpu_baseline={"error": f"{type(e).__name__}: {e}"})
        em.emit()                                           # the headline line: on stdout from here on, whatever comes next
        if selfcheck is not None and selfcheck.get("mismatches"):
            self.finish()
            raise SystemExit(f"bench.py self-check FAILED: {selfcheck['mismatches']} of {selfcheck['elements']} elements of the "
                             f"timed launch differ from the oracle; first: {selfcheck['first_mismatch']}")
        fatal = self.world > 1
        if env_rand and n1 and not a.no_sustained and getattr(self.run, "info", None):
            guarded(em, wd, "sustained", self.sustained)
        if env_rand and n1 and not a.no_single_step:
            guarded(em, wd, "single_step", self.single_step)
        if env_rand and self.distributed and not a.no_gather:
            guarded(em, wd, "control", lambda: self.control_block(self.env, self.N, self.F, self.units_per_step,
                                                                  total_arg=a.total_lanes or None), bound=2 * a.phase_timeout,
                    fatal=fatal)
        if env_rand and n1 and not a.no_secondary:
            def sec():
                s = self.secondary()
                if self.rank == 0 and not a.no_cpu_baseline:
                    try:
                        s["cpu_baseline"] = cpu_baseline_mpc(budget_s=5.0)
                    except Exception as e:                  # noqa: BLE001
                        s["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
                return s
            guarded(em, wd, "secondary", sec, bound=2 * a.phase_timeout)
        if env_rand and n1 and not a.no_mpc_rollout:
            guarded(em, wd, "mpc_rollout", self.mpc_rollout, bound=2 * a.phase_timeout)
        if env_rand and not a.total_lanes and not a.no_strong and STRONG_TOTAL % self.world == 0:
            guarded(em, wd, "strong_1048576", self.strong, bound=3 * a.phase_timeout, fatal=fatal)
        self.finish()

    def finish(self):
        if self.distributed:
            try:
                with self.wd.phase("destroy_process_group", 30.0):
                    self.dist.destroy_process_group()
            except Exception:                               # noqa: BLE001
                pass
        self.wd.stop()


def main(argv=None):
    Run(parse_args(argv)).main()


if __name__ == "__main__":
    main()
