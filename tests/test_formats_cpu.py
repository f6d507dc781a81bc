"""File formats of the reference (8f rank 2): the trace loader is the reference's
own (one float per line); the MPD loader is the repaired intent of a parser that
raises TypeError in the reference (D4)."""
import numpy as np
import pytest


def test_trace_file_roundtrip_is_bit_exact(tmp_path):
    import abrsimulator_amd as A
    rng = np.random.default_rng(0)
    vals = list(rng.uniform(0.1, 9.0, 500)) + [1.0, 0.1, 1e-3, 123456.789, float(np.float32(3.3))]
    p = str(tmp_path / "t.txt")
    A.save_trace_file(p, vals)
    back = A.load_trace_file(p)
    assert back == [float(v) for v in vals]                 # exact float64 equality
    # what the reference does: float(line) per line (Simulator.py:62-63)
    assert back == [float(line) for line in open(p).readlines()]
    net = A.load_network_info(0.5, [p, p])
    assert net.interval == 0.5 and len(net.bandwidths) == 2
    with pytest.raises(ValueError):
        open(str(tmp_path / "e.txt"), "w").close()
        A.load_trace_file(str(tmp_path / "e.txt"))


def test_mpd_file(tmp_path):
    import abrsimulator_amd as A
    lads = [[0.3, 0.75, 1.2], [0.31, 0.8, 1.25], [0.29, 0.7, 1.1]]
    p = str(tmp_path / "v.mpd")
    A.save_mpd_file(p, lads)
    mpd = A.load_mpd_file(4, 20, 8, p)
    assert mpd.video_length == 3 and mpd.chunk_length == 4 and mpd.max_buffer == 20
    assert [list(c.bitrates) for c in mpd.chunks] == lads
    assert list(mpd.chunks[1].sizes) == [b * 4 for b in lads[1]]
    # the lines differ: there is no single ladder, and asking for one is an error, not chunk 0's
    assert not mpd.uniform() and mpd.bitrate_table() == lads
    with pytest.raises(ValueError):
        mpd.ladder()
    A.save_mpd_file(p, [lads[0]] * 3)
    same = A.load_mpd_file(4, 20, 8, p)
    assert same.uniform() and same.ladder() == lads[0]
    assert A.MPD(3, 4, 20, 8, A.Chunk(lads[0])).ladder() == lads[0]
    open(p, "a").write("1 2\n")
    with pytest.raises(ValueError):
        A.load_mpd_file(4, 20, 8, p)


def test_pack_traces_rejects_bad_bandwidths():
    import torch
    from abrsimulator_amd.env import pack_traces
    flat, off, lens = pack_traces([[1.0, 2.0], [3.0]], "cpu")
    assert flat.tolist() == [1.0, 2.0, 3.0] and off.tolist() == [0, 2] and lens.tolist() == [2, 1]
    for bad in ([[1.0, float("nan")]], [[float("inf")]], [[-0.5, 1.0]], [[]], []):
        with pytest.raises(ValueError):
            pack_traces(bad, "cpu")
