#!/bin/bash
# async pipeline iteration loop: parity tests, role stats, same-box A/B vs the role-split kernel
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_async_gpu.py -x -q --timeout 240 > gpurun_out/r03/async_tests.log 2>&1 || { tail -15 gpurun_out/r03/async_tests.log; exit 1; }
tail -2 gpurun_out/r03/async_tests.log
ABR_HIP_LIB=libabr_hip_astats.so timeout -k 10 200 python tools/gpu_async_stats.py 65536 48 || exit 1
for impl in async split async split; do
  timeout -k 10 200 python bench.py --impl $impl --no-cpu-baseline --no-secondary --steps 960 --warmup 96 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['impl'], d['config']['fuse'], '%.3e'%d['value'], '%.1f us/launch'%d['roofline']['avg_launch_us'])" || exit 1
done
