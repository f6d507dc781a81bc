"""Pins the CPU oracle (oracle/abr_oracle.c) to outputs of the REFERENCE itself.

The fixtures in tests/golden/ were produced by tools/gen_golden.py, which runs
the reference's mpc.py as shipped and its Simulator.run() under the R1-R3
control-flow repair.  Everything is compared bit-for-bit
(float64 ==), which is stricter than the 1e-5 relative bar of BASELINE.json.
"""
import json
import os

import numpy as np
import pytest

from conftest import ENV_GOLDENS, GOLDEN, MPC_GOLDENS, load_golden


def _env_cfg(o, m):
    return o.env_cfg(m["ladder"], m["chunk_length"], m["video_length"], m["max_buffer"],
                     m["start_up_length"], m["interval"], m["weights"], m["speed"])


@pytest.mark.parametrize("name", ENV_GOLDENS)
def test_env_episode_bit_exact(oracle, name):
    m, g = load_golden(name)
    cfg = _env_cfg(oracle, m)
    steps, bw, fin, ticks = oracle.env_batch(cfg, list(g["traces"]), g["trace_id"], g["offset"],
                                             g["actions"])
    for k in ["global_time", "rebuffer_time", "start_up_time", "play_time", "average_latency",
              "buffer_level", "play_length", "instant_latency"]:
        assert np.array_equal(steps[k], g[k]), k
    for k in ["chunk_id", "play_id", "start_up", "buffer_empty", "buffer_full"]:
        assert np.array_equal(steps[k], g[k]), k
    assert np.array_equal(steps["last_bitrate"], g["arg_last_bitrate"])
    assert np.array_equal(steps["last_bandwidth"], g["arg_last_bandwidth"])
    assert np.array_equal(bw, g["final_bandwidths"])
    for k in ["qoe", "rebuffer_time", "start_up_time", "average_latency", "global_time",
              "buffer_level", "play_time"]:
        assert np.array_equal(fin[k], g["final_" + k]), k
    assert np.array_equal(fin["play_id"], g["final_play_id"])
    assert ticks > 0


def test_env_speed_schedule_bit_exact(oracle):
    """A scripted speed controller: get_next_speed() answers differently at every played chunk
    (Simulator.py:176-177).  The fixture is the reference itself driven by that script."""
    m, g = load_golden("env_speed_schedule")
    cfg = _env_cfg(oracle, m)
    steps, bw, fin, _ = oracle.env_batch(cfg, list(g["traces"]), g["trace_id"], g["offset"],
                                         g["actions"], speeds=g["speed_sched"])
    for k in ["global_time", "rebuffer_time", "start_up_time", "play_time", "average_latency",
              "buffer_level", "play_length", "instant_latency"]:
        assert np.array_equal(steps[k], g[k]), k
    for k in ["chunk_id", "play_id", "start_up", "buffer_empty", "buffer_full"]:
        assert np.array_equal(steps[k], g[k]), k
    assert np.array_equal(bw, g["final_bandwidths"])
    for k in ["qoe", "rebuffer_time", "start_up_time", "average_latency", "global_time",
              "buffer_level", "play_time"]:
        assert np.array_equal(fin[k], g["final_" + k]), k
    assert np.array_equal(fin["play_id"], g["final_play_id"])
    # the script really was consumed past its end (the last answer repeats) and speeds differ
    assert g["final_speed_calls"].min() > g["speed_sched"].shape[1]
    assert len(np.unique(g["speed_sched"])) > 8


def test_env_golden_exercises_the_edges():
    """The fixtures must actually contain the regimes they are named for."""
    _, g = load_golden("env_bufferfull_i05")
    # chunk 2 could only start once the buffer had drained back under max_buffer=3.0
    # (buffer_full gating, Simulator.py:144,190-191): it starts within two ticks of that
    gated = (g["buffer_level"] < 3.0) & (g["buffer_level"] > 2.97)
    assert gated[:, 2].sum() >= 4 and gated.sum() >= 50
    _, g = load_golden("env_starved_i03")
    assert (g["final_rebuffer_time"] > 1.0).all()
    _, g = load_golden("env_bench_shape")
    # first ABR call happens at tick 401 of the drifting clock, not at 4.00
    assert g["global_time"][0, 0] == 4.009999999999959
    assert g["start_up_time"][0, 0] == 4.019999999999959


def test_mpc_known_answer(oracle):
    """mpc_test.py:52-72,81-86 -> 'Test next bitrate: 2'."""
    with open(os.path.join(GOLDEN, "mpc_known_answer.json")) as f:
        k = json.load(f)
    J_ref = np.load(os.path.join(GOLDEN, "mpc_known_answer_J.npz"))["Jout"]
    assert k["action"] == 2 and k["argmin"] == [2, 1, 3, 3, 3]
    assert k["Jmin"] == -117.56833333333331
    B, V, H = len(k["ladder"]), k["video_length"], k["horizon"]
    br = np.tile(np.array(k["ladder"], np.float64), (V, 1))
    cfg = oracle.mpc_cfg(B, H, V, k["chunk_length"], k["max_buffer"], k["weights"]["variance"],
                         k["weights"]["rebuffer"], k["weights"]["startup"])
    pred, hist = oracle.mpc_predict_list(H, k["history"])
    assert pred.tolist() == k["pred"] and len(hist) == k["hist_len_after"] == 10
    flat, Jmin, J = oracle.mpc_brute(cfg, br, br, k["chunk"], k["prev_bitrate"], k["buffer"], pred)
    assert flat == k["flat"] == 639 and Jmin == k["Jmin"]
    assert np.array_equal(J, J_ref)
    assert J[0] == k["J_first"] and J[-1] == k["J_last"]
    # batched (n, S) form
    hn = np.array([5.0]); hs = np.array([0.0])
    s = 0
    for x in k["history"]:
        s += 1 / x
    hs[0] = s
    act, fl, Jm, pr = oracle.mpc_select(cfg, br, br, [k["chunk"]], [k["prev_bitrate"]],
                                        [k["buffer"]], hn, hs)
    assert act[0] == 2 and fl[0] == 639 and Jm[0] == k["Jmin"] and pr[0].tolist() == k["pred"]
    assert hn[0] == 10.0


@pytest.mark.parametrize("name", MPC_GOLDENS)
def test_mpc_sweep_bit_exact(oracle, name):
    m, g = load_golden(name)
    cfg = oracle.mpc_cfg(m["n_rates"], m["horizon"], m["video_length"], m["chunk_length"],
                         m["max_buffer"], m["variance_weight"], m["rebuffer_weight"],
                         m["startup_weight"])
    hn = g["hist_n"].astype(np.float64)
    hs = g["hist_s"].copy()
    act, flat, Jm, pred = oracle.mpc_select(cfg, g["br"], g["sz"], g["chunk"], g["prev"], g["buf"],
                                            hn, hs)
    assert np.array_equal(pred, g["pred"])
    assert np.array_equal(act, g["action"])
    assert np.array_equal(flat, g["flat"])
    assert np.array_equal(Jm, g["Jmin"])
    assert np.array_equal(hn, g["hist_n_after"].astype(np.float64))
    assert np.array_equal(hs, g["hist_s_after"])
    # literal list-form predictor agrees with the (n, S) form
    for i in range(min(8, len(act))):
        p, h = oracle.mpc_predict_list(m["horizon"], g["hist_raw"][i, :g["hist_n"][i]])
        assert np.array_equal(p, g["pred"][i])
    # full objective grids (near-ties visible)
    for i in range(m["n_full"]):
        f, Jmin, J = oracle.mpc_brute(cfg, g["br"], g["sz"], g["chunk"][i], g["prev"][i],
                                      g["buf"][i], g["pred"][i])
        assert np.array_equal(J, g["Jfull"][i])
        assert f == g["flat"][i]


@pytest.mark.parametrize("name", ["env_bench_shape", "env_starved_i03", "env_speed125", "env_l3_i07"])
def test_pure_python_restatement_bit_exact(name):
    """oracle/pyloop.py (the interpreter-baseline twin) against the same fixtures."""
    from oracle.pyloop import PyTickEnv, run_episode
    m, g = load_golden(name)
    for i in range(0, g["actions"].shape[0], 5):
        env = PyTickEnv(m["ladder"], m["chunk_length"], m["video_length"], m["max_buffer"],
                        m["start_up_length"], m["interval"], m["weights"],
                        g["traces"][g["trace_id"][i]], int(g["offset"][i]), m["speed"])
        obs, qoe = run_episode(env, [int(a) for a in g["actions"][i]])
        assert qoe == g["final_qoe"][i]
        for s, o in enumerate(obs):
            for k in ["global_time", "rebuffer_time", "start_up_time", "play_time",
                      "average_latency", "buffer_level"]:
                assert o[k] == g[k][i, s], (i, s, k)
            assert o["last_bandwidth"] == g["arg_last_bandwidth"][i, s]
        assert env.t == g["final_global_time"][i]


@pytest.mark.parametrize("name", ENV_GOLDENS + ["env_speed_schedule"])
def test_reward_split_of_the_goldens_adds_up_to_the_references_qoe(name):
    """tests/helpers.py: golden_rewards() is the expected value of every per-step reward the GPU tests
    compare element by element.  Pin the derivation itself to the reference: in float64 the split of
    calculate_qoe (Simulator.py:79-86) at the call sites plus the latency term IS run()'s return value."""
    from helpers import golden_rewards
    m, g = load_golden(name)
    r = golden_rewards(m, g, dtype=np.float64)
    assert r.shape == g["actions"].shape
    total = r.sum(1) + m["weights"][3] * g["final_average_latency"]
    assert np.allclose(total, g["final_qoe"], rtol=1e-12, atol=1e-12), np.abs(total - g["final_qoe"]).max()
    # the variance term alone, as calculate_qoe forms it (:81-82)
    lad = np.asarray(m["ladder"])
    var = np.abs(np.diff(lad[g["actions"]], axis=1)).sum(1)
    rest = m["weights"][0] * g["final_rebuffer_time"] + m["weights"][2] * g["final_start_up_time"]
    assert np.allclose(r.sum(1), rest + m["weights"][1] * var, rtol=1e-12, atol=1e-12)
