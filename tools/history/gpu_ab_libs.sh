# Same-box, interleaved A/B of several builds of the library (boxes differ by several percent on issue-bound
# kernels, so only same-box numbers compare).
# usage: bash tools/gpu_ab_libs.sh "<lib> <lib> ..." [rounds] [extra bench args]     ("-" = the product library)
LIBS=${1:-"- libabr_hip_ab.so"}
ROUNDS=${2:-3}
shift 2
for r in $(seq $ROUNDS); do
  for LIB in $LIBS; do
    if [ "$LIB" = "-" ]; then unset ABR_HIP_LIB; else export ABR_HIP_LIB=$LIB; fi
    python bench.py --no-cpu-baseline --no-secondary --no-strong "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s'%'$LIB', '%.4g'%d['value'], '%.1f us/launch'%d['roofline']['avg_launch_us'], 'fuse', d['config'].get('fuse'))"
  done
done
