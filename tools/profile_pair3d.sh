# PMC passes over the 3-D two-step kernel (FWI_STREAM_PAIR=1) next to the single-step stream kernel, 256^3 forward.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ppair
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in pair single; do
  if [ $mode = pair ]; then export FWI_STREAM_PAIR=1; else export FWI_STREAM_PAIR=0; fi
  CMD="python3 $R/bench.py --leg headline --steps 1 --warmup 0 --nt 100"
  pass() { local n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$mode/$n -- $CMD > $O/${mode}_$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/${mode}_$n.log; }; }
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
  pass grbm GRBM_GUI_ACTIVE
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode/kt -- $CMD > $O/${mode}_kt.log 2>&1
  python3 $R/tools/summarize_pmc.py $O/$mode step3d_ > $O/summary_$mode.json
done
cd $R
cat $O/summary_pair.json $O/summary_single.json
