"""Shared helpers for the parity tests (tests may use the oracle; the product may not)."""
import numpy as np


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


def make_env(meta, traces, n_lanes, device="cuda", **kw):
    import abrsimulator_amd as A
    mpd = A.MPD(meta["video_length"], meta["chunk_length"], meta["max_buffer"],
                meta["start_up_length"], A.Chunk(meta["ladder"]))
    qoe = A.QOEMetric(*meta["weights"])
    net = A.NetworkInfo(meta["interval"], [np.asarray(t, np.float64) for t in traces])
    return A.BatchedABREnv(mpd, qoe, net, n_lanes, device=device, speed=meta.get("speed", 1.0), **kw)


F64_EXACT = ["global_time", "rebuffer_time", "start_up_time", "play_time", "buffer_level",
             "play_length"]
