# Round-2 measurement sweep on the GPU box: tests, bench JSON lines, sweeps, rocprofv3 kernel
# stats, HBM traffic counters (FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes), SQ counters,
# per-role cycle stamps.  Everything lands under gpurun_out/r02/; tools/collect_profiles.py
# copies the summaries into profiles/.   usage: gpurun -- bash tools/gpu_r02_final.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_default.err
python bench.py --workload mpc --steps 30 --warmup 5 > $O/bench_mpc.json 2>> $O/bench_default.err
python bench.py --workload env_mpc --steps 96 --warmup 8 --no-cpu-baseline > $O/bench_env_mpc.json 2>> $O/bench_default.err
echo "bench lines done"
rm -f $O/sweeps.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'impl',d['config']['impl'],'fuse',d['config']['fuse'],'lanes',d['config']['lanes_per_gpu'],'value %.4g'%d['value'],'launch_us %.1f'%d['roofline']['avg_launch_us'],'frac %.4f'%d['roofline']['frac'])" | tee -a $O/sweeps.txt; }
for F in 1 16 48; do python bench.py --steps 960 --warmup 96 --fuse $F --no-cpu-baseline --no-secondary 2>/dev/null | line fuse; done
for N in 131072 262144 1048576; do python bench.py --steps 480 --warmup 96 --lanes-per-gpu $N --no-cpu-baseline --no-secondary 2>/dev/null | line lanes; done
python bench.py --mixed-traces --steps 960 --warmup 96 --no-cpu-baseline --no-secondary 2>/dev/null | line mixed_traces_300_3000
for I in jump tick; do python bench.py --impl $I --steps 480 --warmup 96 --no-cpu-baseline --no-secondary 2>/dev/null | line other_impl; done
python bench.py --impl jump --lanes-per-gpu 1048576 --steps 480 --warmup 96 --no-cpu-baseline --no-secondary 2>/dev/null | line other_impl
echo "sweeps done"
make -C abrsimulator_amd/csrc -s libabr_hip_stamps.so      # diagnostic build (not built by __graft_entry__.build)
ABR_HIP_LIB=libabr_hip_stamps.so python tools/gpu_stamps.py 65536 > $O/role_stamps.txt 2>&1 || true
tail -3 $O/role_stamps.txt
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-secondary --steps 960 --warmup 96"
M="$R/bench.py --no-cpu-baseline --workload mpc --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_env -- python3 $B > $O/stats_env.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mpc -- python3 $M > $O/stats_mpc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_env_mpc -- python3 $R/bench.py --no-cpu-baseline --workload env_mpc --steps 48 --warmup 8 > $O/stats_env_mpc.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_env -- python3 $B > $O/fetch_env.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_env -- python3 $B > $O/write_env.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_mpc -- python3 $M > $O/fetch_mpc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_mpc -- python3 $M > $O/write_mpc.log 2>&1
echo "traffic done"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/sq_env_g$i -- python3 $B > $O/sq_env_g$i.log 2>&1 || echo "SQ group $i failed"
done
echo "sq done"
cd $R
python tools/collect_profiles.py --stage $O
