# quick loop: env parity + role stamps + bench at 65 536 lanes (and optionally more sizes)
set -e
mkdir -p gpurun_out/quick
timeout -k 10 600 python -m pytest tests/test_env_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q > gpurun_out/quick/pytest.txt 2>&1 || { tail -40 gpurun_out/quick/pytest.txt; exit 1; }
tail -2 gpurun_out/quick/pytest.txt
ABR_HIP_LIB=libabr_hip_stamps.so timeout -k 10 300 python tools/gpu_stamps.py 65536 | tail -1
for L in ${SIZES:-65536}; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --lanes-per-gpu $L --steps 960 --warmup 96 > gpurun_out/quick/bench_$L.json 2>gpurun_out/quick/bench_$L.err
  python -c "
import json; d=json.loads(open('gpurun_out/quick/bench_$L.json').read().strip().splitlines()[-1]); print($L, '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
done
