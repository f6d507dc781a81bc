# A/B of an environment variable on the same box: bash ab_env.sh VAR valA valB [impl] [lanes]
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-28s %-7s %8d lanes  %.4g env-steps/s  %.1f us/launch  fuse %d' % (sys.argv[1], d['config']['impl'], d['config']['lanes_per_gpu'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))" $1; }
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
for r in 1 2 3; do for v in $2 $3; do env $1=$v timeout -k 10 200 python bench.py --impl ${4:-auto} --lanes-per-gpu ${5:-65536} --steps 960 --warmup 96 $S 2>/dev/null | line "$1=$v"; done; done
for r in 1 2 3; do for v in $2 $3; do env $1=$v timeout -k 10 200 python bench.py --impl ${4:-auto} --lanes-per-gpu ${5:-65536} --steps 20 --warmup 5 $S 2>/dev/null | line "$1=$v"; done; done
