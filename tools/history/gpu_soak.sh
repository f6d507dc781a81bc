# parity soaks at scale (GPU box): every lane against the oracle on the host cores
mkdir -p gpurun_out/r02
S=gpurun_out/r02/soak.jsonl; rm -f $S
python tools/soak_parity.py 1048576 uniform split | tee -a $S
python tools/soak_parity.py 1048576 uniform auto | tee -a $S
python tools/soak_parity.py 262144 mixed split | tee -a $S
python tools/soak_mpc.py 131072 | tee -a $S
python tools/soak_rollout.py 8192 | tee -a $S
python -c "import __graft_entry__ as g; g.smoke()" | tee -a $S
