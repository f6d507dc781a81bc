set -e
python -m pytest tests/test_mpc_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do
python bench.py --workload mpc --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mpc value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'])"
done
