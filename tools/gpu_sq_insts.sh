# Instruction counts of the env kernel for one or more builds of the library (SQ counters, one rocprofv3 --pmc pass each).
# usage: bash tools/gpu_sq_insts.sh "<lib> <lib> ..." outdir [bench args]      ("-" = the product library)
LIBS=$1; O=$GRAFT_REPO_ROOT/$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for LIB in $LIBS; do
  if [ "$LIB" = "-" ]; then unset ABR_HIP_LIB; T=product; else export ABR_HIP_LIB=$LIB; T=$LIB; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $O/sq_$T -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-strong --steps 480 --warmup 48 "$@" > $O/sq_$T.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/sq_$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "env_" in k and "<2>" in k:
        print("$T", k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
