"""Shared helpers for the parity tests (tests may use the oracle; the product may not)."""
import numpy as np


from oracle.oracle import philox_action  # noqa: E402,F401  (bit-exact numpy twin of the device policy)


def make_env(meta, traces, n_lanes, device="cuda", **kw):
    import abrsimulator_amd as A
    mpd = A.MPD(meta["video_length"], meta["chunk_length"], meta["max_buffer"],
                meta["start_up_length"], A.Chunk(meta["ladder"]))
    qoe = A.QOEMetric(*meta["weights"])
    net = A.NetworkInfo(meta["interval"], [np.asarray(t, np.float64) for t in traces])
    return A.BatchedABREnv(mpd, qoe, net, n_lanes, device=device, speed=meta.get("speed", 1.0), **kw)


F64_EXACT = ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level",
             "play_length"]
