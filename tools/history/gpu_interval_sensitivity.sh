# How much of the env kernel's time is the trace-interval changes of a download?  The same bench line with longer trace
# intervals (ABR_BENCH_INTERVAL: a diagnostic override; the workload is then NOT BASELINE's): a download crosses fewer
# interval ends, so the download wave runs fewer trips while the player's work per decision stays about the same.
S="--no-cpu-baseline --no-secondary --no-strong --no-sustained --no-mpc-rollout --no-single-step --steps 1920 --warmup 192"
for r in 1 2; do for I in 1.0 2.0 4.0 1000.0; do
  ABR_BENCH_INTERVAL=$I timeout -k 10 120 python bench.py --allow-overrides $S 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('interval $I s  %.4g env-steps/s  %.1f us/launch' % (d['value'], d['roofline']['avg_launch_us']))"
done; done
for I in 1.0 4.0 1000.0; do echo "interval $I s"; ABR_HIP_LIB=libabr_hip_stamps.so ABR_BENCH_INTERVAL=$I timeout -k 10 200 python tools/gpu_stamps.py 65536 split3 2>/dev/null | grep -E "^\s+\[" ; done
