# round 5: the ring-coupled three-wave kernel (impl ring3) against the barrier form (split3), same box, interleaved
# usage: bash tools/gpu_ab_ring.sh [skip-tests]
O=gpurun_out/r05; mkdir -p $O
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-8s %.4g env-steps/s  %.1f us/launch  fuse %d' % (d['config']['impl'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))"; }
if [ "$1" != "skip-tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "ring3 or FUSED or rollout or timeout or degenerate" > $O/ab_ring_tests.log 2>&1 || { tail -30 $O/ab_ring_tests.log; exit 1; }
  tail -3 $O/ab_ring_tests.log
fi
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
for r in 1 2 3; do
  for I in split3 ring3; do timeout -k 10 120 python bench.py --impl $I --steps 1920 --warmup 192 $S 2>/dev/null | line; done
done 2>&1 | tee $O/ab_ring_fuse48.txt
for r in 1 2 3; do
  for I in split3 ring3; do timeout -k 10 120 python bench.py --impl $I --steps 20 --warmup 5 $S 2>/dev/null | line; done
done 2>&1 | tee $O/ab_ring_fuse20.txt
