cd $GRAFT_REPO_ROOT
tools/ab_try.sh nb tools/lib_nb6.so tools/lib_nb2.so > gpurun_out/ab_nb.log 2>&1
