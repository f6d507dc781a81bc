set -e
mkdir -p gpurun_out/r02c
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02c/pytest.txt 2>&1 || { tail -40 gpurun_out/r02c/pytest.txt; exit 1; }
tail -3 gpurun_out/r02c/pytest.txt
for L in 65536 262144 1048576; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --lanes-per-gpu $L --steps 960 --warmup 96 > gpurun_out/r02c/bench_$L.json 2>gpurun_out/r02c/bench_$L.err
  python -c "
import json; d=json.loads(open('gpurun_out/r02c/bench_$L.json').read().strip().splitlines()[-1]); print($L, '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
done
