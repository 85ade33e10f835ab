# PMC passes over the 2-D fused kernel (forward / SAVE_Q / IMAGE) on configs[1] (1024^2, O(8) + sponge).
# SQ block: 8 counters per pass; TCC and GRBM separately.  Summaries: tools/summarize_pmc.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/p2d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/time_config.py --config cfg2 --nt 400 --rounds 1"
pass() { local n=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $CMD > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/$n.log; }; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass grbm GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $CMD > $O/kt.log 2>&1
cd $R
python3 tools/summarize_pmc.py $O step2d_fused > $O/summary.json && cat $O/summary.json
