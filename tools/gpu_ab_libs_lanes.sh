# same-box interleaved A/B of library builds at a given lane count:  bash tools/gpu_ab_libs_lanes.sh "libA.so libB.so" impl lanes [rounds] [steps]
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s %-7s %8d lanes  %.4g env-steps/s  %.1f us/launch  fuse %d' % (sys.argv[1], d['config']['impl'], d['config']['lanes_per_gpu'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))" $1; }
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
for r in $(seq 1 ${4:-3}); do for L in $1; do ABR_HIP_LIB=$L timeout -k 10 200 python bench.py --impl $2 --lanes-per-gpu $3 --steps ${5:-960} --warmup 96 $S 2>/dev/null | line $L; done; done
