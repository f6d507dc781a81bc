# Collects the round's rocprofv3 evidence on the GPU box: kernel-trace stats of the default
# bench command, and HBM traffic counters (FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes,
# as MI355X_MICROARCH.md prescribes).  usage: bash tools/gpu_profile.sh <tag> [bench args]
set -e
tag=${1:-r01}; shift || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/write.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
def avg(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
out = {"FETCH_SIZE_KB": avg(O + "/fetch", "FETCH_SIZE"), "WRITE_SIZE_KB": avg(O + "/write", "WRITE_SIZE")}
stats = {}
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0]] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
out["kernel_stats"] = stats
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
