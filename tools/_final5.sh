set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1100 python tools/fuzz_soak.py 24 401 41 > gpurun_out/r04/soak41.log 2>&1 || { tail -20 gpurun_out/r04/soak41.log; echo SOAK_FAILED; }
tail -2 gpurun_out/r04/soak41.log
