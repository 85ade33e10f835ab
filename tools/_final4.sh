set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -q > gpurun_out/r04/round4.log 2>&1 || tail -30 gpurun_out/r04/round4.log
tail -2 gpurun_out/r04/round4.log
timeout -k 10 1000 bash tools/collect_r04_runs.sh > gpurun_out/r04/runs.log 2>&1 || { tail -20 gpurun_out/r04/runs.log; echo RUNS_FAILED; }
grep -E "^==|seconds" gpurun_out/r04/runs.log | cut -c1-300
