# Issue / wait accounting of the env kernel (SQ counters, one rocprofv3 --pmc pass per group): how busy the vector issue is,
# how many lanes are active per vector instruction, how much of a wave's life is waiting.
#   usage: [KERNEL=substring] bash tools/gpu_sq_util.sh outdir [bench args]      (KERNEL defaults to the env kernels' `<2>` instances)
O=$GRAFT_REPO_ROOT/$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step --steps 480 --warmup 48 "$@" > $O/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        want = "${KERNEL:-}"
        if (want and want in k) or (not want and "env_" in k and "<2>" in k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, json.dumps({c: round(v) for c, v in m.items()}))
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        print("  lanes active per VALU instruction: %.1f" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] * 1.0))
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
        print("  wait fraction of wave-cycles: %.3f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
    if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
        # GRBM_GUI_ACTIVE is reported SUMMED over the 8 XCDs (round 4 divided by the sum and printed 0.109 for K3 where the
        # kernel fills 0.90 of its issue slots): kernel cycles = GRBM_GUI_ACTIVE / 8
        print("  VALU issue utilisation (4 cycles x instr / SIMD / kernel cycles, kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs): %.3f"
              % (4 * m["SQ_INSTS_VALU"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8.0)))
PY
