set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
rm -f gpurun_out/quick.log
for F in 1 16 48; do
  python bench.py --steps 480 --warmup 96 --fuse $F --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse',d['config']['fuse'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'],'frac %.4f'%d['roofline']['frac'])" | tee -a gpurun_out/quick.log
done
