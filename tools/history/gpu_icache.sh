R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05/icache; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step --steps 480 --warmup 48"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/g1 -- python3 $R/bench.py $LEAN > $O/g1.log 2>&1 || echo g1 failed
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/g2 -- python3 $R/bench.py $LEAN > $O/g2.log 2>&1 || echo g2 failed
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $O/g3 -- python3 $R/bench.py $LEAN > $O/g3.log 2>&1 || echo g3 failed
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "env_split3" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
tail -3 $O/g1.log
