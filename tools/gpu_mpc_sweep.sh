for L in 16 14 7 5 3 1; do
ABR_MPC_LPB=$L python bench.py --workload mpc --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('lpb',os.environ.get('ABR_MPC_LPB'),'mpc value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'])"
done
