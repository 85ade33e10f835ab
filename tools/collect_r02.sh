# Round-2 evidence: GPU tests, the bench line, rocprofv3 kernel stats per leg, FETCH/WRITE PMC passes per leg.
# Run on the GPU box:  bash tools/collect_r02.sh   (writes gpurun_out/r02/, summaries are copied to profiles/ by hand)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
prof() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" > $O/$name.log 2>&1 || { echo "rocprofv3 $name failed"; tail -5 $O/$name.log; return 1; }
}
prof kt_headline --kernel-trace --stats --output-format csv -d $O/kt_headline -- python3 $R/bench.py --leg headline --steps 3 --warmup 1 &&
prof kt_hbm --kernel-trace --stats --output-format csv -d $O/kt_hbm -- python3 $R/bench.py --leg hbm &&
prof kt_gradient --kernel-trace --stats --output-format csv -d $O/kt_gradient -- python3 $R/bench.py --leg gradient &&
prof kt_cfg2 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python3 $R/bench.py --leg cfg2 &&
for c in FETCH_SIZE WRITE_SIZE; do
  prof pmc_headline_$c --pmc $c --output-format csv -d $O/pmc_headline_$c -- python3 $R/bench.py --leg headline --steps 1 --warmup 0 --nt 100 &&
  prof pmc_hbm_$c --pmc $c --output-format csv -d $O/pmc_hbm_$c -- python3 $R/bench.py --leg hbm --leg-nt 30 &&
  prof pmc_gradient_$c --pmc $c --output-format csv -d $O/pmc_gradient_$c -- python3 $R/bench.py --leg gradient --leg-nt 100 &&
  prof pmc_cfg2_$c --pmc $c --output-format csv -d $O/pmc_cfg2_$c -- python3 $R/bench.py --leg cfg2 --leg-nt 200 || exit 1
done
cd $R
find $O -name "*.csv" | sort
