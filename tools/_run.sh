cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak4
WRSN_K=8000 timeout -k 10 300 python tools/soak.py > gpurun_out/soak4/n200.log 2>&1 &&
WRSN_K=2000 WRSN_N=1000 WRSN_M=8 timeout -k 10 400 python tools/soak.py > gpurun_out/soak4/n1000.log 2>&1 &&
WRSN_K=1500 WRSN_N=200 WRSN_M=8 timeout -k 10 300 python tools/soak.py > gpurun_out/soak4/n200_m8.log 2>&1
