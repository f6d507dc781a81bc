# which implementation wins where (same box): lanes x {split, split3, jump}, fused 48 and single steps
mkdir -p gpurun_out/r04
S="--no-cpu-baseline --no-secondary --no-strong"
for N in 16384 65536 98304 131072 196608 262144; do
  for I in split split3 jump; do
    python bench.py --impl $I --lanes-per-gpu $N --steps 960 --warmup 96 $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $N', '$I', 'fuse', d['config']['fuse'], '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
for N in 65536 131072; do
  for I in split split3 jump; do
    python bench.py --impl $I --lanes-per-gpu $N --fuse 1 --steps 960 --warmup 96 $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $N', '$I', 'fuse', d['config']['fuse'], '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
for I in split split3; do python bench.py --impl $I --steps 20 --warmup 5 $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-args', '$I', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"; done
for I in split split3; do python bench.py --impl $I --workload env_mpc --steps 96 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('env_mpc', '$I', '%.4g'%d['value'])"; done
