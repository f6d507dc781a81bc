# LDS bank conflicts of the env kernel's mailboxes and parking areas (one rocprofv3 --pmc pass):  bash tools/gpu_lds_conflicts.sh outdir
O=$GRAFT_REPO_ROOT/$1; R=$GRAFT_REPO_ROOT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step --steps 480 --warmup 48 > $O/lds.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "env_split3" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: round(v) for c, v in m.items()})
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m and m["SQ_ACTIVE_INST_LDS"]:
        print("  bank-conflict cycles / active LDS cycles: %.4f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_ACTIVE_INST_LDS"]))
PY
