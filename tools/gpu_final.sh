set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_final.json 2>gpurun_out/bench_final.err
python bench.py --workload mpc --steps 30 --warmup 5 > gpurun_out/bench_final_mpc.json 2>>gpurun_out/bench_final.err
python bench.py --workload env_mpc --steps 96 --warmup 8 --no-cpu-baseline > gpurun_out/bench_final_env_mpc.json 2>>gpurun_out/bench_final.err
rm -f gpurun_out/sweep_final.log
for F in 1 16 48; do python bench.py --steps 960 --warmup 96 --fuse $F --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse',d['config']['fuse'],'lanes',d['config']['lanes_per_gpu'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'],'frac %.4f'%d['roofline']['frac'])" | tee -a gpurun_out/sweep_final.log; done
for N in 131072 262144 1048576; do python bench.py --steps 480 --warmup 96 --fuse 48 --lanes-per-gpu $N --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse',d['config']['fuse'],'lanes',d['config']['lanes_per_gpu'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'],'frac %.4f'%d['roofline']['frac'])" | tee -a gpurun_out/sweep_final.log; done
python bench.py --mixed-traces --steps 960 --warmup 96 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mixed traces value %.4g'%d['value'])" | tee -a gpurun_out/sweep_final.log
python - <<'PY' | tee -a gpurun_out/sweep_final.log
import sys, time, torch
sys.path.insert(0, '.')
import bench, abrsimulator_amd as A
traces = bench.synth_traces()
N = 65536
tid, off = bench.lane_assignment(0, N, traces)
for impl in ("jump", "tick"):
    env = A.BatchedABREnv(A.MPD(48, 4.0, 20.0, 8.0, A.Chunk(bench.LADDER)), A.QOEMetric(*bench.WEIGHTS),
                          A.NetworkInfo(1.0, traces), N, auto_reset=True, impl=impl)
    env.reset(torch.from_numpy(tid), torch.from_numpy(off))
    env.step_random(48, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): env.step_random(48, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("impl", impl, "env-steps/s %.4g" % (N * 48 * 5 / dt))
PY
bash tools/gpu_profile.sh r01b_env
bash tools/gpu_profile.sh r01b_mpc --workload mpc --steps 20 --warmup 3
bash tools/gpu_pmc.sh r01b --steps 480 --warmup 96 2>&1 | grep "kernel<2>" || true
