# same-box interleaved A/B of product-library builds: bash tools/gpu_ab_libs2.sh "libA.so libB.so" impl [rounds]
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s %-7s %.4g env-steps/s  %.1f us/launch  fuse %d' % (sys.argv[1], d['config']['impl'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))" $1; }
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
for r in $(seq 1 ${3:-3}); do for L in $1; do ABR_HIP_LIB=$L timeout -k 10 120 python bench.py --impl $2 --steps 1920 --warmup 192 $S 2>/dev/null | line $L; done; done
for r in $(seq 1 ${3:-3}); do for L in $1; do ABR_HIP_LIB=$L timeout -k 10 120 python bench.py --impl $2 --steps 20 --warmup 5 $S 2>/dev/null | line $L; done; done
