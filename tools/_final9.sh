set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -rf gpurun_out/r04/kt_* gpurun_out/r04/pmc_*
timeout -k 10 1150 bash tools/collect_r04.sh all > gpurun_out/r04/collect_all.log 2>&1 || { tail -20 gpurun_out/r04/collect_all.log; exit 1; }
tail -3 gpurun_out/r04/collect_all.log
