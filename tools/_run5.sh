set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
rm -f gpurun_out/r04e/*
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x > gpurun_out/r04e/t1.log 2>&1 || tail -40 gpurun_out/r04e/t1.log
tail -2 gpurun_out/r04e/t1.log
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04e/t.log 2>&1
echo "== VL 4 lines" >> gpurun_out/r04e/t.log
FWI_PML_LINE_VL=4 timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04e/t.log 2>&1
cat gpurun_out/r04e/t.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04e/kt_cpml3d -- python3 $GRAFT_REPO_ROOT/bench.py --leg cpml3d > $GRAFT_REPO_ROOT/gpurun_out/r04e/kt.log 2>&1 || tail -5 $GRAFT_REPO_ROOT/gpurun_out/r04e/kt.log
cd $GRAFT_REPO_ROOT
find gpurun_out/r04e/kt_cpml3d -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
