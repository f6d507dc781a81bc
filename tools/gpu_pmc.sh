# usage: bash tools/gpu_pmc.sh <tag> <bench args...>   (one rocprofv3 --pmc pass per counter group)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag/g$i -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/pmc_$tag/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ["GRAFT_REPO_ROOT"]; 
import sys
for d in sorted(glob.glob(R+"/gpurun_out/pmc_*/g*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "env_" in k or "mpc_" in k:
                print(d.split("gpurun_out/")[1], k, {c:(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
