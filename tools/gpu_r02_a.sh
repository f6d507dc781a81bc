# round 2, call A: instruction cost model + regression of the hardened ABI + baseline numbers
set -e
mkdir -p gpurun_out/r02a
./tools/microbench/valu_cost > gpurun_out/r02a/valu_cost.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.txt 2>&1 || { tail -30 gpurun_out/r02a/pytest.txt; exit 1; }
tail -3 gpurun_out/r02a/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2>gpurun_out/r02a/bench_default.err
python bench.py --no-cpu-baseline --lanes-per-gpu 262144 --steps 960 --warmup 96 > gpurun_out/r02a/bench_262144.json 2>/dev/null
cat gpurun_out/r02a/valu_cost.txt
python - <<'PY'
import json
for f in ("bench_default","bench_262144"):
    d=json.loads(open(f"gpurun_out/r02a/{f}.json").read().strip().splitlines()[-1])
    print(f, "%.4g"%d["value"], d["roofline"]["avg_launch_us"])
PY
