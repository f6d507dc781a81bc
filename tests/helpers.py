"""Shared helpers for the parity tests (tests may use the oracle; the product may not)."""
import os
import subprocess

import numpy as np


from oracle.oracle import philox_action  # noqa: E402,F401  (bit-exact numpy twin of the device policy)


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_IMPLS = ("async", "ring3", "pair3")      # pipelines that were measured slower and live in the diagnostic build only


_diag_lib_cache = {}


def diag_lib():
    """tools/diag/lib/libabr_hip_diag.so -- the product's translation unit plus the rejected pipelines (impl 'async',
    'ring3', 'pair3') -- built on demand (hipcc, ~40 s) when missing or older than its sources, ONCE per test session.
    Tests name it explicitly (BatchedABREnv(library=...)); the product package never loads it by itself.  Without hipcc
    a missing or stale library skips the test that asked for it; a compile error fails it with the compiler's output."""
    import shutil

    import pytest
    if "so" in _diag_lib_cache:
        return _diag_lib_cache["so"]
    d = os.path.join(_ROOT, "tools", "diag", "csrc")
    so = os.path.join(_ROOT, "tools", "diag", "lib", "libabr_hip_diag.so")
    srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    c = os.path.join(_ROOT, "abrsimulator_amd", "csrc")
    srcs += [os.path.join(c, f) for f in os.listdir(c) if f.endswith((".h", ".hip"))]
    srcs.append(os.path.join(_ROOT, "include", "abr_env.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not (os.path.exists(hipcc) or shutil.which("hipcc")):
            pytest.skip("the diagnostic library (rejected pipelines) is missing or older than its sources and there is no "
                        "hipcc to build it: make -C tools/diag/csrc")
        r = subprocess.run(["make", "-C", d, "-s", "../lib/libabr_hip_diag.so"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("building tools/diag/lib/libabr_hip_diag.so failed (the rejected pipelines no longer compile "
                               "against the product's headers?):\n" + r.stdout[-4000:])
    _diag_lib_cache["so"] = so
    return so


def make_env(meta, traces, n_lanes, device="cuda", **kw):
    import abrsimulator_amd as A
    if kw.get("impl") in DIAG_IMPLS and "library" not in kw:
        kw["library"] = diag_lib()
    mpd = A.MPD(meta["video_length"], meta["chunk_length"], meta["max_buffer"],
                meta["start_up_length"], A.Chunk(meta["ladder"]))
    qoe = A.QOEMetric(*meta["weights"])
    net = A.NetworkInfo(meta["interval"], [np.asarray(t, np.float64) for t in traces])
    return A.BatchedABREnv(mpd, qoe, net, n_lanes, device=device, speed=meta.get("speed", 1.0), **kw)


F64_EXACT = ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level",
             "play_length"]


def expected_rewards(*args, **kw):
    """Per-step linear QoE reward derived from REFERENCE quantities, float32 [N, V]: oracle.step_rewards (it lives with
    the oracle so that bench.py's self-check of the timed launch can use the same derivation)."""
    from oracle.oracle import step_rewards
    return step_rewards(*args, **kw)


def golden_rewards(meta, g, dtype=np.float32):
    """expected_rewards() from one of the reference's golden fixtures (tests/golden/env_*)."""
    return expected_rewards(g["rebuffer_time"], g["start_up_time"], g["final_rebuffer_time"],
                            g["final_start_up_time"], g["actions"], meta["weights"], ladder=meta["ladder"],
                            dtype=dtype)


def oracle_rewards(steps, fin, actions, weights, ladder=None, br_table=None):
    """expected_rewards() from the oracle's per-step records (oracle.env_batch / env_batch_mpc)."""
    return expected_rewards(steps["rebuffer_time"], steps["start_up_time"], fin["rebuffer_time"],
                            fin["start_up_time"], actions, weights, ladder=ladder, br_table=br_table)


def auto_reset_rollout_rewards(oracle, cfg, traces, trace_id, offset, actions, weights, ladder=None,
                               br_table=None, **kw):
    """Expected float32 rewards [n_steps, N] of a fused rollout under auto_reset whose every lane ends its
    episodes after exactly V decisions (no time-outs): `actions` [n_steps, N] as the kernel reported them.
    Each episode is replayed through the oracle from the lane's (trace, offset); a partial last episode is
    padded with bitrate 0 (a step's reward depends on the actions up to that step only)."""
    actions = np.asarray(actions)
    n_steps, N = actions.shape
    V = cfg.video_length
    out = np.zeros((n_steps, N), np.float32)
    for e0 in range(0, n_steps, V):
        n = min(V, n_steps - e0)
        a = np.zeros((N, V), np.int32)
        a[:, :n] = actions[e0:e0 + n].T
        steps, _, fin, _ = oracle.env_batch(cfg, traces, trace_id, offset, a, **kw)
        out[e0:e0 + n] = oracle_rewards(steps, fin, a, weights, ladder=ladder, br_table=br_table)[:, :n].T
    return out
