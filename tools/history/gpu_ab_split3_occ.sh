S="--no-cpu-baseline --no-secondary --no-strong"
for r in 1 2; do
for N in 98304 131072; do
  for V in "A:split" "A:split3" "libabr_hip_ab_occ5.so:split3" "libabr_hip_ab_occ6.so:split3"; do
    L=${V%%:*}; I=${V#*:}
    if [ $L = A ]; then unset ABR_HIP_LIB; else export ABR_HIP_LIB=$L; fi
    python bench.py --impl $I --lanes-per-gpu $N --steps 960 --warmup 96 $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $N', '$L', '$I', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
done
