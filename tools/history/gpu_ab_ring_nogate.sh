line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-22s %-7s %.4g env-steps/s  %.1f us/launch  fuse %d' % (sys.argv[1], d['config']['impl'], d['value'], d['roofline']['avg_launch_us'], d['config']['fuse']))" $1; }
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step"
export ABR_BENCH_MAX_BUFFER=1e9
for r in 1 2 3; do
  ABR_HIP_LIB=libabr_hip.so timeout -k 10 120 python bench.py --allow-overrides --impl split3 --steps 1920 --warmup 192 $S 2>/dev/null | line libabr_hip.so
  for L in libabr_hip.so libabr_hip_v3.so libabr_hip_v8.so; do ABR_HIP_LIB=$L timeout -k 10 120 python bench.py --allow-overrides --impl ring3 --steps 1920 --warmup 192 $S 2>/dev/null | line $L; done
done
for r in 1 2; do
  ABR_HIP_LIB=libabr_hip.so timeout -k 10 120 python bench.py --allow-overrides --impl split3 --steps 20 --warmup 5 $S 2>/dev/null | line libabr_hip.so
  for L in libabr_hip_v3.so libabr_hip_v8.so; do ABR_HIP_LIB=$L timeout -k 10 120 python bench.py --allow-overrides --impl ring3 --steps 20 --warmup 5 $S 2>/dev/null | line $L; done
done
