cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 &&
WRSN_N=1000 WRSN_M=8 WRSN_B=48 WRSN_K=40 WRSN_BUDGET=1250 timeout -k 10 600 python tests/parity_sweep.py > gpurun_out/sweep_rr_1000.log 2>&1 &&
WRSN_B=192 WRSN_K=60 WRSN_BUDGET=1250 timeout -k 10 300 python tests/parity_sweep.py > gpurun_out/sweep_rr_200.log 2>&1 &&
WRSN_B=192 WRSN_K=40 WRSN_SEED=31000 timeout -k 10 300 python tests/parity_sweep.py > gpurun_out/sweep_rr_200_blocking.log 2>&1
