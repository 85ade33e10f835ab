set -o pipefail
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/mcp; mkdir -p $R/gpurun_out/mcp
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mcp/kt -- python3 $R/tests/bench_mc.py --cpu-samples 2 > $R/gpurun_out/mcp/kt.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/mcp/pmc -- python3 $R/tests/bench_mc.py --cpu-samples 2 > $R/gpurun_out/mcp/pmc.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/mcp/pmc2 -- python3 $R/tests/bench_mc.py --cpu-samples 2 > $R/gpurun_out/mcp/pmc2.log 2>&1
find $R/gpurun_out/mcp -name "*.csv" | head -20
