# rocprofv3 kernel stats + SQ counters of ONE variant of the env kernel (e.g. the one-thread-per-lane
# kernel at a size where the role-split kernel is the default), for profiles/ side by side with the kept one.
# usage: bash tools/gpu_profile_variant.sh <tag> <bench args...>
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03/variant_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-secondary --steps 960 --warmup 96 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/stats.log 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/sq_g$i -- python3 $B > $O/sq_g$i.log 2>&1 || echo "SQ group $i failed"
done
cd $R
python3 - "$O" "$tag" <<'PY'
import collections, csv, glob, json, sys
O, tag = sys.argv[1], sys.argv[2]
c, kern = {}, None
for g in ("sq_g1", "sq_g2", "sq_g3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + "/" + g + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("env_") and k.endswith("<2>"):
            kern = k
            c.update({n: sum(x) / len(x) for n, x in v.items()})
stats = {}
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0].replace("void ", "")] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]))
LANES, FUSE = 65536, 48
d = {"valu_insts_per_lane_group_per_decision": c["SQ_INSTS_VALU"] / (LANES / 64) / FUSE,
     "salu_insts_per_lane_group_per_decision": c["SQ_INSTS_SALU"] / (LANES / 64) / FUSE,
     "lds_insts_per_lane_group_per_decision": c["SQ_INSTS_LDS"] / (LANES / 64) / FUSE,
     "vmem_insts_per_lane_group_per_decision": (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / (LANES / 64) / FUSE,
     "waves": c["SQ_WAVES"],
     "wave_cycles_per_decision": c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"] / FUSE,
     "active_inst_any_fraction": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
     "wait_inst_any_fraction": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
     "wait_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
     "avg_active_lanes_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
     "valu_issue_frac_x4": c["SQ_INSTS_VALU"] / 1024 * 4.0 / (stats[kern]["avg_ns"] * 1e-9 * 2.4e9)}
json.dump({"variant": tag, "kernel": kern, "lanes": LANES, "fuse": FUSE, "kernel_stats": stats[kern],
           "per_launch_average": c, "derived": d}, open(O + "/summary.json", "w"), indent=1)
print(json.dumps({"kernel": kern, "stats": stats[kern], "derived": d}, indent=1))
PY
