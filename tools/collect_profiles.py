#!/usr/bin/env python3
"""Summarise the round's evidence from gpurun_out/<round>/ (scratch, written by tools/gpu_r0N_final.sh on
the GPU box) into profiles/ (tracked): rocprofv3 kernel stats, PMC summaries, bench JSON lines.

  python tools/collect_profiles.py --stage DIR   on the GPU box: DIR/*/.../*.csv -> DIR/summary.json
  python tools/collect_profiles.py               here: gpurun_out/<round>/ -> profiles/<round>_*      (round = ABR_ROUND_TAG, default r06)
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("ABR_ROUND_TAG", "r06")
ENV_KERNEL = "env_split3_kernel<2>"
MPC_KERNEL = "mpc_select_kernel<5, 6, 1>"
LANES, FUSE = 65536, 48


def counters(path):
    """{kernel: {counter: (mean, n)}} over every dispatch of a --pmc run."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


def stats(path):
    out = {}
    for f in glob.glob(path + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"].split("(")[0].replace("void ", "")] = dict(
                calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
    return out


ENV_RUNS = (("env", 48), ("env_f20", 20))     # (directory tag, decisions per launch): default bench, driver's --steps 20


def stage(O):
    s = {"stats_mpc": stats(O + "/stats_mpc"), "stats_env_mpc": stats(O + "/stats_env_mpc")}
    for tag, _ in ENV_RUNS:
        s["stats_" + tag] = stats(O + "/stats_" + tag)
        for k in ("fetch_", "write_"):
            s[k + tag] = counters(O + "/" + k + tag)
        for g in (1, 2, 3):
            s[f"sq_{tag}_g{g}"] = counters(O + f"/sq_{tag}_g{g}")
    for k in ("fetch_mpc", "write_mpc"):
        s[k] = counters(O + "/" + k)
    json.dump(s, open(O + "/summary.json", "w"), indent=1)
    print(json.dumps({k: list(v)[:4] for k, v in s.items()}, indent=1)[:2500])


def collect():
    G, P = os.path.join(R, "gpurun_out", TAG), os.path.join(R, "profiles")
    s = json.load(open(os.path.join(G, "summary.json")))
    for src in ("bench_default.json", "bench_driver_args.json", "bench_mpc.json", "bench_env_mpc.json", "bench_env_mpc_mixed.json",
                "sweeps.txt", "sweep_impl.txt", "role_stamps_split.txt", "role_stamps_split3.txt", "role_stamps_split_131072.txt",
                "role_stamps_ring3.txt", "async_role_stats.txt", "placement_split3.txt",
                "mpc_phase_stamps.txt", "mpc_sq_counters.txt", "gpu_fuzz.json", "gpu_fuzz_extended.json"):
        if not os.path.exists(os.path.join(G, src)):
            continue
        if src.startswith("bench_"):
            # bench.py prints its line again after every block: the LAST line is the complete one, and it is what is kept
            lines = [ln for ln in open(os.path.join(G, src)).read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(P, f"{TAG}_{src}"), "w").write(lines[-1] + "\n")
            continue
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{TAG}_{src}"))
    if os.path.exists(os.path.join(G, "soak_extended.jsonl")):
        shutil.copy(os.path.join(G, "soak_extended.jsonl"), os.path.join(P, f"{TAG}_soak_parity_extended.jsonl"))
    for name, dst in (("stats_env", "env_random_fuse48_kernel_stats.csv"), ("stats_env_f20", "env_random_fuse20_kernel_stats.csv"),
                      ("stats_mpc", "mpc_kernel_stats.csv"), ("stats_env_mpc", "env_mpc_kernel_stats.csv")):
        fs = sorted(glob.glob(os.path.join(G, name, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # (earlier runs of the round leave theirs behind)
        if fs:
            shutil.copy(fs[-1], os.path.join(P, f"{TAG}_{dst}"))
    if os.path.exists(os.path.join(G, "soak.jsonl")):
        shutil.copy(os.path.join(G, "soak.jsonl"), os.path.join(P, f"{TAG}_soak_parity.jsonl"))
    keys = [k + t for t, _ in ENV_RUNS for k in ("fetch_", "write_")] + ["fetch_mpc", "write_mpc"]
    json.dump({k: s[k] for k in keys}, open(os.path.join(P, f"{TAG}_hbm_pmc_summary.json"), "w"), indent=1)
    # ---- HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE doubled on gfx950, WRITE_SIZE as read) ----
    src = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_r0N_final.sh), "
           "per-launch averages in KB; FETCH_SIZE doubled (gfx950 reports half of a coalesced read, "
           "MI355X_MICROARCH.md 'HBM'; our 4-8 B/lane reads are outside the calibrated 16 B/lane case, "
           "so the read side is an upper estimate), WRITE_SIZE as read")
    traffic = {"env_random": []}

    def entry(kern, f, w, fuse, pick):
        fk = [k for k in s[f] if pick(k)]
        if not fk:
            return None
        k = fk[0]
        fe, nf = s[f][k]["FETCH_SIZE"]
        wr, _ = s[w][k]["WRITE_SIZE"]
        return {"fuse": fuse, "lanes": LANES, "kernel": kern, "profiled_kernel": k, "launches": nf,
                "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "bytes_per_launch": (2 * fe + wr) * 1024,
                "source": f"profiles/{TAG}_hbm_pmc_summary.json ({f}, {w}): " + src}
    for tag, fuse in ENV_RUNS:
        e = entry(ENV_KERNEL, "fetch_" + tag, "write_" + tag, fuse, lambda k: k.startswith(ENV_KERNEL))
        if e:
            traffic["env_random"].append(e)
    e = entry("mpc_select_kernel<5,6>", "fetch_mpc", "write_mpc", 1, lambda k: k.startswith("mpc_select_kernel"))
    if e:
        traffic["mpc"] = e
    json.dump(traffic, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
    # ---- SQ counters of the env kernel and what binds it, per profiled fuse ----
    for tag, fuse in ENV_RUNS:
        c = {}
        for g in (1, 2, 3):
            for k, v in s[f"sq_{tag}_g{g}"].items():
                if k.startswith(ENV_KERNEL):
                    c.update({n: m for n, (m, _) in v.items()})
        ks = [v for k, v in s["stats_" + tag].items() if k.startswith(ENV_KERNEL)]
        if not (c and ks):
            continue
        waves, n_simd, dur_s = c["SQ_WAVES"], 1024, ks[0]["avg_ns"] * 1e-9
        valu_per_simd = c["SQ_INSTS_VALU"] / n_simd
        derived = {
            "waves_per_launch": waves,
            "valu_insts_per_lane_group_per_decision": c["SQ_INSTS_VALU"] / (LANES / 64) / fuse,
            "salu_insts_per_lane_group_per_decision": c["SQ_INSTS_SALU"] / (LANES / 64) / fuse,
            "lds_insts_per_lane_group_per_decision": c["SQ_INSTS_LDS"] / (LANES / 64) / fuse,
            "wave_cycles_per_decision": c["SQ_WAVE_CYCLES"] * 4 / waves / fuse,
            "wait_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "active_fraction": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "avg_active_lanes_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]}
        binding = {"resource": "valu_issue (one wave's dependent float64 stream per SIMD pair: see DESIGN.md section 4)",
                   "frac": valu_per_simd * 4.0 / (dur_s * 2.4e9),
                   "active_lanes": derived["avg_active_lanes_per_valu_inst"],
                   "valu_insts_per_simd_per_launch": valu_per_simd, "kernel_us": dur_s * 1e6,
                   "definition": "vector instructions per SIMD per launch x 4 cycles / (kernel time x 2.4 GHz); "
                                 "active_lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)"}
        json.dump({"kernel": ENV_KERNEL, "lanes": LANES, "fuse": fuse,
                   "config": f"{LANES} lanes, fuse {fuse}, three waves (download, player, service roles) per 64 lanes",
                   "per_launch_average": c, "derived": derived, "binding": binding},
                  open(os.path.join(P, f"{TAG}_env_split3_fuse{fuse}_sq_counters.json"), "w"), indent=1)
        print("binding fuse", fuse, binding)
    print("traffic", {k: (v["bytes_per_launch"] if isinstance(v, dict) else [(e["fuse"], e["bytes_per_launch"]) for e in v])
                      for k, v in traffic.items()})
    print("kernel stats", {t: {k: v for k, v in s["stats_" + t].items() if "env_" in k} for t, _ in ENV_RUNS})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="")
    a = ap.parse_args()
    stage(a.stage) if a.stage else collect()
