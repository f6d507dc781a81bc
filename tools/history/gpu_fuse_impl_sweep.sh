# decisions per launch x implementation (and lanes x implementation at ONE decision per launch): which kernel `auto`
# should take for the stepwise API.   usage (GPU box): bash tools/gpu_fuse_impl_sweep.sh
S="--no-cpu-baseline --no-secondary --no-strong"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes',d['config']['lanes_per_gpu'],'fuse',d['config']['fuse'],'impl',d['config']['impl'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'], 'ms_per_step %.5f'%d['ms_per_step'])"; }
for F in 1 2 4 8; do for I in split3 split jump; do python bench.py --steps 960 --warmup 96 --fuse $F --impl $I $S 2>/dev/null | line; done; done
for N in 4096 16384 32768 131072; do for I in split3 split jump; do python bench.py --steps 960 --warmup 96 --fuse 1 --impl $I --lanes-per-gpu $N $S 2>/dev/null | line; done; done
