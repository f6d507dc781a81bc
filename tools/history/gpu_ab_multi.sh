# several B variants against A on the same box.  usage: bash tools/gpu_ab_multi.sh lanes lib1.so lib2.so ...
L=$1; shift
for r in 1 2 3; do
  for V in A "$@"; do
    if [ $V = A ]; then unset ABR_HIP_LIB; else export ABR_HIP_LIB=$V; fi
    python bench.py --no-cpu-baseline --no-secondary --lanes-per-gpu $L --steps 1920 --warmup 192 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
