#!/usr/bin/env python3
"""Copy the round's evidence from gpurun_out/ (scratch, written by tools/gpu_final.sh on the
GPU box) into profiles/ (tracked): rocprofv3 kernel stats, PMC summaries, bench JSON lines."""
import collections
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
TAG = "r01"
F = 48

def cp(src_glob, dst):
    src = sorted(glob.glob(os.path.join(G, src_glob)))[-1]
    shutil.copy(src, os.path.join(P, dst))

cp("profile_r01b_env/stats/*/*_kernel_stats.csv", f"{TAG}_env_random_fuse{F}_kernel_stats.csv")
cp("profile_r01b_mpc/stats/*/*_kernel_stats.csv", f"{TAG}_mpc_kernel_stats.csv")
cp("profile_r01b_env/summary.json", f"{TAG}_env_random_fuse{F}_pmc_summary.json")
cp("profile_r01b_mpc/summary.json", f"{TAG}_mpc_pmc_summary.json")
cp("bench_final.json", f"{TAG}_bench_default.json")
cp("bench_final_mpc.json", f"{TAG}_bench_mpc.json")
cp("bench_final_env_mpc.json", f"{TAG}_bench_env_mpc.json")
cp("sweep_final.log", f"{TAG}_sweeps.txt")

out = {}
for d in sorted(glob.glob(os.path.join(G, "pmc_r01b/g*/"))):
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1:]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "env_jump_kernel<2>" in k:
                out.update({c: sum(x) / len(x) for c, x in v.items()})
json.dump({"kernel": "env_jump_kernel<2>",
           "config": f"65536 lanes, fuse {F} (one launch = {F} decisions per lane), 1024 waves",
           "per_launch_average": out,
           "derived": {
               "valu_insts_per_wave_per_decision": out["SQ_INSTS_VALU"] / 1024 / F,
               "salu_insts_per_wave_per_decision": out["SQ_INSTS_SALU"] / 1024 / F,
               "vmem_rd_insts_per_wave_per_decision": out["SQ_INSTS_VMEM_RD"] / 1024 / F,
               "wave_cycles_per_decision": out["SQ_WAVE_CYCLES"] * 4 / 1024 / F,
               "wait_fraction": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"],
               "active_fraction": out["SQ_ACTIVE_INST_ANY"] / out["SQ_WAVE_CYCLES"],
               "avg_active_lanes_per_valu_inst": out["SQ_THREAD_CYCLES_VALU"] / out["SQ_ACTIVE_INST_VALU"]}},
          open(os.path.join(P, f"{TAG}_env_jump_sq_counters.json"), "w"), indent=1)

d = json.load(open(os.path.join(G, "profile_r01b_env/summary.json")))
f, nf = d["FETCH_SIZE_KB"]["void env_jump_kernel<2>"]
w, nw = d["WRITE_SIZE_KB"]["void env_jump_kernel<2>"]
m = json.load(open(os.path.join(G, "profile_r01b_mpc/summary.json")))
km = [k for k in m["FETCH_SIZE_KB"] if "mpc_select" in k][0]
fm, wm = m["FETCH_SIZE_KB"][km][0], m["WRITE_SIZE_KB"][km][0]
src = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_profile.sh), "
       "per-launch averages; FETCH_SIZE doubled (gfx950 reports half of a coalesced read, "
       "MI355X_MICROARCH.md 'HBM'; our 4-8 B/lane reads are outside the calibrated 16 B/lane case, "
       "so the read side is an upper estimate), WRITE_SIZE as read")
json.dump({"env_random": {"fuse": F, "lanes": 65536, "kernel": "env_jump_kernel<2>", "launches": nf,
                          "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "bytes_per_launch": (2 * f + w) * 1024,
                          "source": f"profiles/{TAG}_env_random_fuse{F}_pmc_summary.json: " + src},
           "mpc": {"fuse": 1, "lanes": 65536, "kernel": km, "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
                   "bytes_per_launch": (2 * fm + wm) * 1024,
                   "source": f"profiles/{TAG}_mpc_pmc_summary.json: " + src}},
          open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
ks = d["kernel_stats"]["void env_jump_kernel<2>"]
b = json.loads(open(os.path.join(G, "bench_final.json")).read().strip().split("\n")[-1])
print("rocprof avg_ns", ks["avg_ns"], "calls", ks["calls"], "| bench avg_launch_us", b["roofline"]["avg_launch_us"],
      "value", b["value"], "traffic", b["roofline"]["traffic"])
