# What the host-side wait for the GPU costs in the driver's bracket (barrier + synchronize around ONE 20-decision
# launch): the same bench line under different runtime wait policies, same box, interleaved.
#   default                      interrupt-driven wait
#   ROC_ACTIVE_WAIT_TIMEOUT=us   the HIP runtime spins on the completion signal for that long before it blocks
#   HSA_ENABLE_INTERRUPT=0       the HSA runtime polls signals instead of sleeping on an interrupt
for r in 1 2 3; do
  for V in "X=1" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000 HSA_ENABLE_INTERRUPT=0"; do
    env $V python bench.py --no-cpu-baseline --no-secondary --no-strong --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-52s'%'$V', '%.4g'%d['value'], 'ms_per_step %.5f'%d['ms_per_step'], '%.1f us/launch (events)'%d['roofline']['avg_launch_us'])"
  done
done
