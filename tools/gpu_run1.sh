set -e
mkdir -p gpurun_out
for F in 1 4 16 48; do
  python bench.py --steps 480 --warmup 96 --fuse $F --no-cpu-baseline >> gpurun_out/sweep_fuse.log 2>&1
done
python bench.py --workload mpc --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/sweep_fuse.log 2>&1
python bench.py --workload env_mpc --steps 96 --warmup 8 --no-cpu-baseline >> gpurun_out/sweep_fuse.log 2>&1
python bench.py > gpurun_out/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_env -- python3 $GRAFT_REPO_ROOT/bench.py --steps 192 --warmup 48 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_env.log 2>&1
