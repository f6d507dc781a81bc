# parity soaks at scale (GPU box), round 4: every lane against the oracle on the host cores
mkdir -p gpurun_out/r04
S=gpurun_out/r04/soak.jsonl; rm -f $S
python tools/soak_parity.py 1048576 uniform split3 | tee -a $S
python tools/soak_parity.py 1048576 uniform split | tee -a $S
python tools/soak_parity.py 1048576 uniform auto | tee -a $S
python tools/soak_parity.py 1048576 uniform async | tee -a $S
python tools/soak_parity.py 262144 mixed async | tee -a $S
python tools/soak_parity.py 262144 mixed split | tee -a $S
python tools/soak_parity.py 262144 mixed split3 | tee -a $S
python tools/soak_mpc.py 131072 | tee -a $S
python tools/soak_rollout.py 8192 | tee -a $S
python tools/gpu_fuzz.py 1200 512 | tee gpurun_out/r04/gpu_fuzz.json
