# a longer device fuzz and bigger MPC / rollout soaks than tools/gpu_r05_final.sh 3 (GPU box, ~10 minutes; the oracle on the host cores is the clock)
mkdir -p gpurun_out/r05
S=gpurun_out/r05/soak_extended.jsonl; rm -f $S
python tools/soak_mpc.py 524288 | tee -a $S
python tools/soak_rollout.py 32768 | tee -a $S
python tools/gpu_fuzz.py 6000 512 | tee gpurun_out/r05/gpu_fuzz_extended.json
