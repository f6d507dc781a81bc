mkdir -p gpurun_out/r03
ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_stamps.py 65536 split3 2>&1 | grep -v amdgpu | tee gpurun_out/r03/role_stamps_split3.txt
for r in 1 2; do
  for V in A libabr_hip_ab_d1p0.so libabr_hip_ab_d2p1.so libabr_hip_ab_d1p2.so libabr_hip_ab_d2p2.so; do
    if [ $V = A ]; then unset ABR_HIP_LIB; else export ABR_HIP_LIB=$V; fi
    python bench.py --impl split3 --no-cpu-baseline --no-secondary --no-strong --steps 1920 --warmup 192 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', '%.4g'%d['value'], '%.1f us'%d['roofline']['avg_launch_us'])"
  done
done 2>&1 | tee gpurun_out/r03/ab_split3_prio.txt
