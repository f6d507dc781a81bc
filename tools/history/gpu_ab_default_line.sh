for r in 1 2; do for L in libabr_hip_ab_head.so libabr_hip.so; do
ABR_HIP_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-sustained 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); m=d['mpc_rollout']; print('%-22s value %.4g  single_step %.2f us  mpc_rollout %.4g / %.4g  strong1M %.4g  secondary %.4g  selfcheck %s' % (sys.argv[1], d['value'], d['single_step']['us_per_launch'], m['configs2_65536']['value'], m['configs4_rank7_131072_mixed']['value'], d['strong_1048576']['value'], d['secondary']['value'], (d.get('selfcheck') or {}).get('mismatches')))" $L
done; done
