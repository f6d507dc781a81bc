"""ctypes front-end of the CPU oracle (libabr_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under abrsimulator_amd/ may.

Parity status: pinned (see abr_oracle.c header and tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class EnvCfg(C.Structure):
    _fields_ = [("n_rates", C.c_int32), ("video_length", C.c_int32),
                ("chunk_length", C.c_double), ("max_buffer", C.c_double),
                ("start_up_length", C.c_double), ("interval", C.c_double),
                ("rebuffer_weight", C.c_double), ("variance_weight", C.c_double),
                ("startup_weight", C.c_double), ("latency_weight", C.c_double),
                ("speed", C.c_double), ("ladder", C.c_double * 16), ("br_table", C.c_void_p),
                ("speed_sched", C.c_void_p), ("speed_rows", C.c_int32), ("speed_stride", C.c_int64)]


class MpcCfg(C.Structure):
    _fields_ = [("n_rates", C.c_int32), ("horizon", C.c_int32), ("video_length", C.c_int32),
                ("_pad", C.c_int32), ("chunk_length", C.c_double), ("max_buffer", C.c_double),
                ("variance_weight", C.c_double), ("rebuffer_weight", C.c_double),
                ("startup_weight", C.c_double)]


STEP_DTYPE = np.dtype([
    ("global_time", "f8"), ("rebuffer_time", "f8"), ("start_up_time", "f8"), ("play_time", "f8"),
    ("average_latency", "f8"), ("buffer_level", "f8"), ("play_length", "f8"),
    ("instant_latency", "f8"), ("last_bandwidth", "f8"),
    ("chunk_id", "i4"), ("play_id", "i4"), ("last_bitrate", "i4"),
    ("start_up", "i4"), ("buffer_empty", "i4"), ("buffer_full", "i4")], align=True)

FINAL_DTYPE = np.dtype([
    ("qoe", "f8"), ("rebuffer_time", "f8"), ("start_up_time", "f8"), ("average_latency", "f8"),
    ("global_time", "f8"), ("buffer_level", "f8"), ("play_time", "f8"),
    ("ticks", "i8"), ("play_id", "i4"), ("chunk_id", "i4")], align=True)


def build(force=False):
    so = os.path.join(_HERE, "libabr_oracle.so")
    src = os.path.join(_HERE, "abr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libabr_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_env_batch.restype = C.c_int64
        _LIB.oracle_env_batch_speeds.restype = C.c_int64
        _LIB.oracle_env_batch_mpc.restype = C.c_int64
        _LIB.oracle_env_batch_sched.restype = C.c_int64
        _LIB.oracle_mpc_brute.restype = C.c_int64
        _LIB.oracle_mpc_objective.restype = C.c_double
        assert STEP_DTYPE.itemsize == 96 and FINAL_DTYPE.itemsize == 72
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def env_cfg(ladder, chunk_length, video_length, max_buffer, start_up_length, interval,
            weights, speed=1.0, br_table=None):
    """br_table: optional [video_length][n_rates] per-chunk ladders (build-defined VBR
    generalisation; see abr_oracle.c).  The array is kept alive on the returned struct."""
    c = EnvCfg()
    if br_table is not None:
        c._br_keep = np.ascontiguousarray(br_table, np.float64)
        assert c._br_keep.shape == (int(video_length), len(ladder))
        c.br_table = c._br_keep.ctypes.data
    c.n_rates, c.video_length = len(ladder), int(video_length)
    c.chunk_length, c.max_buffer = float(chunk_length), float(max_buffer)
    c.start_up_length, c.interval = float(start_up_length), float(interval)
    (c.rebuffer_weight, c.variance_weight, c.startup_weight, c.latency_weight) = map(float, weights)
    c.speed = float(speed)
    for i, b in enumerate(ladder):
        c.ladder[i] = float(b)
    return c


def pack_traces(traces):
    """list of 1-D arrays (ragged ok) -> (flat f8, off i8, len i4)."""
    lens = np.array([len(t) for t in traces], np.int32)
    off = np.zeros(len(traces), np.int64)
    off[1:] = np.cumsum(lens[:-1])
    flat = np.concatenate([np.asarray(t, np.float64) for t in traces])
    return np.ascontiguousarray(flat), off, lens


def env_batch(cfg, traces, trace_id, offset, actions, max_ticks=1 << 40, speeds=None,
              want_steps=True):
    """Replay episodes. traces: list of arrays. actions: [N, V] int32.  speeds: optional
    per-lane constant play speeds [N] (default: cfg.speed for every lane), or per-lane speed
    SCHEDULES [N, rows]: the answers to each lane's successive get_next_speed() calls.
    Returns (steps[N,V] STEP_DTYPE, bw[N,V], final[N] FINAL_DTYPE, total_ticks)."""
    flat, off, lens = pack_traces(traces)
    trace_id = np.ascontiguousarray(trace_id, np.int32)
    offset = np.ascontiguousarray(offset, np.int32)
    actions = np.ascontiguousarray(actions, np.int32)
    N, V = actions.shape
    assert V == cfg.video_length
    steps = np.zeros((N, V), STEP_DTYPE) if want_steps else None
    bw = np.zeros((N, V), np.float64)
    fin = np.zeros(N, FINAL_DTYPE)
    sp = steps.ctypes.data_as(C.c_void_p) if want_steps else None
    if speeds is not None and np.ndim(speeds) == 2:
        speeds = np.ascontiguousarray(speeds, np.float64)
        assert speeds.shape[0] == N
        rc = lib().oracle_env_batch_sched(
            C.byref(cfg), _p(flat, C.c_double), _p(off, C.c_int64), _p(lens, C.c_int32),
            _p(trace_id, C.c_int32), _p(offset, C.c_int32), _p(actions, C.c_int32),
            _p(speeds, C.c_double), C.c_int32(speeds.shape[1]), C.c_int32(N), sp,
            _p(bw, C.c_double), fin.ctypes.data_as(C.c_void_p), C.c_int64(max_ticks))
    elif speeds is not None:
        speeds = np.ascontiguousarray(speeds, np.float64)
        assert speeds.shape == (N,)
        rc = lib().oracle_env_batch_speeds(
            C.byref(cfg), _p(flat, C.c_double), _p(off, C.c_int64), _p(lens, C.c_int32),
            _p(trace_id, C.c_int32), _p(offset, C.c_int32), _p(actions, C.c_int32),
            _p(speeds, C.c_double), C.c_int32(N), sp,
            _p(bw, C.c_double), fin.ctypes.data_as(C.c_void_p), C.c_int64(max_ticks))
    else:
        rc = lib().oracle_env_batch(
            C.byref(cfg), _p(flat, C.c_double), _p(off, C.c_int64), _p(lens, C.c_int32),
            _p(trace_id, C.c_int32), _p(offset, C.c_int32), _p(actions, C.c_int32), C.c_int32(N),
            sp, _p(bw, C.c_double), fin.ctypes.data_as(C.c_void_p),
            C.c_int64(max_ticks))
    if rc < 0:
        raise RuntimeError(f"oracle_env_batch failed: {rc}")
    return steps, bw, fin, int(rc)


def env_batch_mpc(cfg, mcfg, br, sz, traces, trace_id, offset, max_ticks=1 << 40, threads=1):
    """Episodes of Simulator.run() driven by the MPC (see abr_oracle.c: composition).
    Returns (steps[N,V], bw[N,V], actions[N,V], final[N]).  threads > 1 splits the lanes
    over a thread pool (ctypes releases the GIL)."""
    flat, off, lens = pack_traces(traces)
    br = np.ascontiguousarray(br, np.float64)
    sz = np.ascontiguousarray(sz, np.float64)
    trace_id = np.ascontiguousarray(trace_id, np.int32)
    offset = np.ascontiguousarray(offset, np.int32)
    N, V = len(trace_id), cfg.video_length
    steps = np.zeros((N, V), STEP_DTYPE)
    bw = np.zeros((N, V), np.float64)
    acts = np.zeros((N, V), np.int32)
    fin = np.zeros(N, FINAL_DTYPE)
    L = lib()

    def run(lo, hi):
        if hi <= lo:
            return 0
        rc = L.oracle_env_batch_mpc(
            C.byref(cfg), C.byref(mcfg), _p(br, C.c_double), _p(sz, C.c_double),
            _p(flat, C.c_double), _p(off, C.c_int64), _p(lens, C.c_int32),
            _p(trace_id[lo:hi], C.c_int32), _p(offset[lo:hi], C.c_int32), C.c_int32(hi - lo),
            steps[lo:hi].ctypes.data_as(C.c_void_p), _p(bw[lo:hi], C.c_double),
            _p(acts[lo:hi], C.c_int32), fin[lo:hi].ctypes.data_as(C.c_void_p), C.c_int64(max_ticks))
        if rc < 0:
            raise RuntimeError(f"oracle_env_batch_mpc failed: {rc}")
        return rc

    if threads <= 1:
        run(0, N)
    else:
        from concurrent.futures import ThreadPoolExecutor
        cuts = np.linspace(0, N, threads + 1).astype(int)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda i: run(cuts[i], cuts[i + 1]), range(threads)))
    return steps, bw, acts, fin


POLICY_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


def env_episode_policy(cfg, trace, offset, policy, max_ticks=1 << 40):
    """One episode driven by a Python callback policy(obs_record, prev_bw_array) -> int."""
    trace = np.ascontiguousarray(trace, np.float64)
    V = cfg.video_length
    steps = np.zeros(V, STEP_DTYPE)
    bw = np.zeros(V, np.float64)
    acts = np.zeros(V, np.int32)
    fin = np.zeros(1, FINAL_DTYPE)

    def cb(_ctx, obs_ptr, bw_ptr, n):
        obs = np.frombuffer((C.c_char * STEP_DTYPE.itemsize).from_address(obs_ptr),
                            dtype=STEP_DTYPE, count=1)[0]
        hist = np.ctypeslib.as_array(bw_ptr, shape=(n,)).copy() if n else np.zeros(0)
        return int(policy(obs, hist))

    rc = lib().oracle_env_episode(
        C.byref(cfg), _p(trace, C.c_double), C.c_int32(len(trace)), C.c_int32(int(offset)),
        None, POLICY_FN(cb), None, steps.ctypes.data_as(C.c_void_p), _p(bw, C.c_double),
        _p(acts, C.c_int32), fin.ctypes.data_as(C.c_void_p), C.c_int64(max_ticks))
    if rc:
        raise RuntimeError(f"oracle_env_episode failed: {rc}")
    return steps, bw, acts, fin[0]


def mpc_cfg(n_rates, horizon, video_length, chunk_length, max_buffer, variance_weight,
            rebuffer_weight, startup_weight=0.0):
    c = MpcCfg()
    c.n_rates, c.horizon, c.video_length = int(n_rates), int(horizon), int(video_length)
    c.chunk_length, c.max_buffer = float(chunk_length), float(max_buffer)
    c.variance_weight, c.rebuffer_weight = float(variance_weight), float(rebuffer_weight)
    c.startup_weight = float(startup_weight)
    return c


def mpc_predict_list(horizon, hist):
    """Returns (pred[H], mutated history list) -- mpc.py:81-93 literal."""
    h = np.zeros(len(hist) + horizon, np.float64)
    h[:len(hist)] = hist
    n = C.c_int32(len(hist))
    pred = np.zeros(horizon, np.float64)
    lib().oracle_mpc_predict_list(C.c_int32(horizon), _p(h, C.c_double), C.byref(n),
                                  _p(pred, C.c_double))
    return pred, h[:n.value]


def mpc_predict_ns(horizon, n, S):
    """(n, S)-form predictor: returns (pred[H], n_after, S_after)."""
    cn, cs = C.c_double(float(n)), C.c_double(float(S))
    pred = np.zeros(horizon, np.float64)
    lib().oracle_mpc_predict_ns(C.c_int32(horizon), C.byref(cn), C.byref(cs), _p(pred, C.c_double))
    return pred, cn.value, cs.value


def mpc_brute(cfg, br, sz, chunk, prev, buf, pred, want_J=True):
    br = np.ascontiguousarray(br, np.float64)
    sz = np.ascontiguousarray(sz, np.float64)
    pred = np.ascontiguousarray(pred, np.float64)
    J = np.zeros(cfg.n_rates ** cfg.horizon, np.float64) if want_J else None
    Jmin = C.c_double()
    flat = lib().oracle_mpc_brute(C.byref(cfg), _p(br, C.c_double), _p(sz, C.c_double),
                                  C.c_int(int(chunk)), C.c_int(int(prev)), C.c_double(float(buf)),
                                  _p(pred, C.c_double), _p(J, C.c_double), C.byref(Jmin))
    return int(flat), Jmin.value, J


def mpc_select(cfg, br, sz, chunk, prev, buf, hist_n, hist_s):
    """Batched next_bitrate. hist_n/hist_s are updated IN PLACE (D9).
    Returns (action i4[N], flat i8[N], Jmin f8[N], pred f8[N,H])."""
    br = np.ascontiguousarray(br, np.float64)
    sz = np.ascontiguousarray(sz, np.float64)
    chunk = np.ascontiguousarray(chunk, np.int32)
    prev = np.ascontiguousarray(prev, np.int32)
    buf = np.ascontiguousarray(buf, np.float64)
    assert hist_n.dtype == np.float64 and hist_s.dtype == np.float64
    N = len(chunk)
    act = np.zeros(N, np.int32)
    flat = np.zeros(N, np.int64)
    Jm = np.zeros(N, np.float64)
    pred = np.zeros((N, cfg.horizon), np.float64)
    rc = lib().oracle_mpc_select(C.byref(cfg), _p(br, C.c_double), _p(sz, C.c_double),
                                 _p(chunk, C.c_int32), _p(prev, C.c_int32), _p(buf, C.c_double),
                                 _p(hist_n, C.c_double), _p(hist_s, C.c_double), C.c_int32(N),
                                 _p(act, C.c_int32), _p(flat, C.c_int64), _p(Jm, C.c_double),
                                 _p(pred, C.c_double))
    if rc:
        raise RuntimeError(f"oracle_mpc_select failed: {rc}")
    return act, flat, Jm, pred


def philox_action(seed, lane, step, episode, n_rates):
    """Bit-exact numpy twin of csrc/abr_env.hip: philox_action (philox4x32-10)."""
    lane = np.asarray(lane, np.uint64)
    c0 = (lane & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    c1 = (lane >> np.uint64(32)).astype(np.uint64)
    c2 = np.broadcast_to(np.asarray(step, np.uint64), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(episode, np.uint64), c0.shape).copy()
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return ((c0 * np.uint64(n_rates)) >> np.uint64(32)).astype(np.int32)


def step_rewards(rebuffer_time, start_up_time, final_rebuffer_time, final_start_up_time, actions,
                 weights, ladder=None, br_table=None, dtype=np.float32):
    """Per-step linear QoE reward derived from REFERENCE quantities, float32 [N, V].

    calculate_qoe (Simulator.py:79-86) split at the ABR call sites (:155) with the timers of :137-140:
        r_s = wr * (rebuffer_time[s+1] - rebuffer_time[s]) + ws * (start_up_time[s+1] - start_up_time[s])
              + wv * |br[s][a_s] - br[s-1][a_(s-1)]|
    where index s is the s-th call site's run() frame, s = 0 takes its deltas from 0 (the time before the
    first call site belongs to the first decision) and has no variance term, and the last step takes
    the frame calculate_qoe was called from (final_*).  br[s] is the single ladder, or chunk s's own row
    of a per-chunk table.  Inputs are the goldens' / the oracle's float64 arrays [N, V] and [N]; the
    operation order is the kernels' ((wr*d_rb + ws*d_su) + wv*var in float64, then one rounding to
    float32), so the comparison is `==`."""
    rb = np.asarray(rebuffer_time, np.float64)
    su = np.asarray(start_up_time, np.float64)
    a = np.asarray(actions)
    N, V = a.shape
    wr, wv, ws = float(weights[0]), float(weights[1]), float(weights[2])
    rb_next = np.concatenate([rb[:, 1:], np.asarray(final_rebuffer_time, np.float64)[:, None]], 1)
    su_next = np.concatenate([su[:, 1:], np.asarray(final_start_up_time, np.float64)[:, None]], 1)
    rb_prev = np.concatenate([np.zeros((N, 1)), rb[:, 1:]], 1)
    su_prev = np.concatenate([np.zeros((N, 1)), su[:, 1:]], 1)
    if br_table is not None:
        tab = np.asarray(br_table, np.float64)                       # [V][B]
        br = tab[np.arange(V)[None, :], a]                           # br[s][a_s]
    else:
        br = np.asarray(ladder, np.float64)[a]
    var = np.zeros((N, V))
    var[:, 1:] = np.abs(br[:, 1:] - br[:, :-1])
    r = (wr * (rb_next - rb_prev) + ws * (su_next - su_prev)) + wv * var
    return r.astype(dtype)
