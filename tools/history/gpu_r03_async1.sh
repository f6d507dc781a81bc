#!/bin/bash
# round 3, first GPU contact of the asynchronous pipeline: its parity tests, then same-box A/B vs the role-split kernel
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_async_gpu.py tests/test_env_gpu.py -x -q --timeout 240 > gpurun_out/r03/async_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r03/async_tests.log
[ $rc -ne 0 ] && exit $rc
for impl in async split async split; do
  timeout -k 10 200 python bench.py --impl $impl --no-cpu-baseline --no-secondary --steps 960 --warmup 96 > gpurun_out/r03/bench_${impl}_$RANDOM.json 2>gpurun_out/r03/bench_err.log || { tail -5 gpurun_out/r03/bench_err.log; exit 1; }
done
for f in gpurun_out/r03/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print(d['config']['impl'], d['config']['fuse'], '%.3e'%d['value'], '%.1f us/launch'%d['roofline']['avg_launch_us'])"; done
timeout -k 10 200 python bench.py --impl async --no-cpu-baseline --no-secondary --steps 20 --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps20', d['config']['impl'], '%.3e'%d['value'])"
timeout -k 10 200 python bench.py --impl split --no-cpu-baseline --no-secondary --steps 20 --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps20', d['config']['impl'], '%.3e'%d['value'])"
