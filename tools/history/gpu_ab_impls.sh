# same-box interleaved A/B of IMPLEMENTATIONS: the product's (no library named) and the diagnostic build's
#   usage: bash tools/gpu_ab_impls.sh "split3 pair3 ring3" [rounds]      (split3 / split / jump run on the product library)
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-7s %.4g env-steps/s  %.1f us/launch  fuse %d' % (d['config']['impl'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))"; }
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
run() { case $1 in ring3|pair3|async) ABR_HIP_LIB=libabr_hip_diag.so timeout -k 10 120 python bench.py --impl $1 $2 $S 2>/dev/null | line;; *) timeout -k 10 120 python bench.py --impl $1 $2 $S 2>/dev/null | line;; esac; }
for r in $(seq 1 ${2:-3}); do for I in $1; do run $I "--steps 1920 --warmup 192"; done; done
for r in $(seq 1 ${2:-3}); do for I in $1; do run $I "--steps 20 --warmup 5"; done; done
