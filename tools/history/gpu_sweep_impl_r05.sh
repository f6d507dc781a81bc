line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('impl',d['config']['impl'],'lanes',d['config']['lanes_per_gpu'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'])"; }
LEAN="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
for N in 81920 98304 114688 131072 163840 196608 262144; do for I in split jump; do python bench.py --impl $I --steps 480 --warmup 96 --lanes-per-gpu $N $LEAN 2>/dev/null | line; done; done
for N in 32768 49152 65536; do for I in split3 split jump; do python bench.py --impl $I --steps 480 --warmup 96 --lanes-per-gpu $N $LEAN 2>/dev/null | line; done; done
