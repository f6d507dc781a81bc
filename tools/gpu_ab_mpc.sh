# K3 A/B on the same box, interleaved.  usage: bash tools/gpu_ab_mpc.sh libB.so [libC.so ...]
for r in 1 2 3; do
  for V in A "$@"; do
    if [ $V = A ]; then unset ABR_HIP_LIB; else export ABR_HIP_LIB=$V; fi
    python bench.py --workload mpc --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', '%.4g combos/s'%d['value'], '%.1f us/select (events)'%d['roofline']['avg_launch_us'], '%.1f us wall'%(d['ms_per_step']*1e3))"
  done
done
