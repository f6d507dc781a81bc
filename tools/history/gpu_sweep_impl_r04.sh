# which implementation wins where (same box, round 4 kernels): lanes x {split3, split, jump}, fused 48
S="--no-cpu-baseline --no-secondary --no-strong"
for N in 32768 65536 81920 98304 114688 131072 163840 196608; do
  for I in split3 split jump; do
    python bench.py --impl $I --lanes-per-gpu $N --steps 960 --warmup 96 $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $N', '$I', 'fuse', d['config']['fuse'], '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
