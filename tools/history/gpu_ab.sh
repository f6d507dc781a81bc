# A/B of two builds of the library on the SAME box, interleaved (A B A B ...): medians compared.
# usage: bash tools/gpu_ab.sh [lanes]      (B = csrc/libabr_hip_ab.so, built with AB_FLAGS)
L=${1:-65536}
for r in 1 2 3; do
  for V in A B; do
    if [ $V = B ]; then export ABR_HIP_LIB=libabr_hip_ab.so; else unset ABR_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-secondary --lanes-per-gpu $L --steps 1920 --warmup 192 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done
