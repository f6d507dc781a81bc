# same-box A/B of kernel IMPLEMENTATIONS inside one library build, interleaved.  usage: bash tools/gpu_ab_impl.sh lanes steps implA implB ...
L=$1; K=$2; shift; shift
for r in 1 2 3; do
  for I in "$@"; do
    python bench.py --impl $I --no-cpu-baseline --no-secondary --no-strong --lanes-per-gpu $L --steps $K --warmup $((K/10+5)) 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$I', 'fuse', d['config']['fuse'], '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
