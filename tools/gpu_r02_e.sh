set -e
mkdir -p gpurun_out/r02e
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02e/bench_driver_args.json 2> gpurun_out/r02e/bench_driver_args.err || { tail -20 gpurun_out/r02e/bench_driver_args.err; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/r02e/bench_default.json 2> gpurun_out/r02e/bench_default.err || { tail -20 gpurun_out/r02e/bench_default.err; exit 1; }
ABR_BENCH_ONE_DEVICE=1 ABR_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 96 --warmup 48 > gpurun_out/r02e/bench_2rank_gloo.json 2> gpurun_out/r02e/bench_2rank_gloo.err || { tail -20 gpurun_out/r02e/bench_2rank_gloo.err; exit 1; }
ABR_BENCH_ONE_DEVICE=1 ABR_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 96 --warmup 48 --fuse 1 --total-lanes 131072 > gpurun_out/r02e/bench_2rank_gloo_fuse1.json 2> gpurun_out/r02e/bench_2rank_gloo_fuse1.err || { tail -20 gpurun_out/r02e/bench_2rank_gloo_fuse1.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_driver_args","bench_default","bench_2rank_gloo","bench_2rank_gloo_fuse1"):
    d=json.loads(open(f"gpurun_out/r02e/{f}.json").read().strip().splitlines()[-1])
    print(f, "%.4g"%d["value"], d["config"]["fuse"], d.get("repeats"), "%.1f"%d["roofline"]["avg_launch_us"], "%.4f"%d["roofline"]["frac"],
          ("sec %.4g"%d["secondary"]["value"]) if d.get("secondary") else "", d["config"]["collective"][:60])
PY
