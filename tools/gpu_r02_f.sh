set -e
mkdir -p gpurun_out/r02f
timeout -k 10 900 python -m pytest tests/test_mpc_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > gpurun_out/r02f/pytest.txt 2>&1 || { tail -40 gpurun_out/r02f/pytest.txt; exit 1; }
tail -2 gpurun_out/r02f/pytest.txt
for W in mpc env_mpc; do
  timeout -k 10 300 python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02f/bench_$W.json 2> gpurun_out/r02f/bench_$W.err || { tail gpurun_out/r02f/bench_$W.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r02f/bench_$W.json').read().strip().splitlines()[-1]); print('$W', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'], d['ms_per_step'])"
done
