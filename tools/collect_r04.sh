# Round-4 evidence: the bench line, rocprofv3 kernel stats per leg, FETCH/WRITE PMC passes per leg -- now also for the
# option kernels DESIGN.md quotes (increment form, bf16 store, fp64, point kernel, CPML in 2-D and 3-D).
# Run on the GPU box:  bash tools/collect_r04.sh [part]   (part 1: bench + kernel stats; part 2: PMC passes)
# Writes gpurun_out/r04/; tools/make_r04_profiles.py turns that into profiles/r04_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
PART=${1:-all}
mkdir -p $O
cd $R
LEGS=${LEGS:-"headline hbm gradient gradient_increment cfg2 cfg2_cpml cpml3d cpml3d_adjoint cpml512 fp64 point bf16"}
cd /tmp && export TMPDIR=/tmp
prof() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  rm -rf $O/$name
  timeout -k 10 300 rocprofv3 "$@" > $O/$name.log 2>&1 || { echo "rocprofv3 $name failed"; tail -5 $O/$name.log; return 1; }
  echo "done $name"
}
args_of() {  # leg -> bench.py arguments of the kernel-stats pass
  case $1 in
    headline) echo "--leg headline --steps 3 --warmup 1";;
    *) echo "--leg $1";;
  esac
}
pmc_args_of() {  # shorter runs for the counter passes (every dispatch is serialised under --pmc)
  case $1 in
    headline) echo "--leg headline --steps 1 --warmup 0 --nt 100";;
    hbm) echo "--leg hbm --leg-nt 30";;
    cfg2|cfg2_cpml) echo "--leg $1 --leg-nt 200";;
    cpml3d|cpml3d_adjoint) echo "--leg $1 --leg-nt 40";;
    cpml512) echo "--leg cpml512 --leg-nt 12";;
    *) echo "--leg $1 --leg-nt 100";;
  esac
}
if [ $PART = all ] || [ $PART = 1 ]; then
  (cd $R && python bench.py > $O/bench.json 2> $O/bench.err) || { tail -5 $O/bench.err; exit 1; }
  cut -c1-300 $O/bench.json
  for leg in $LEGS; do
    prof kt_$leg --kernel-trace --stats --output-format csv -d $O/kt_$leg -- python3 $R/bench.py $(args_of $leg) || exit 1
  done
fi
if [ $PART = all ] || [ $PART = 2 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    for leg in $LEGS; do
      prof pmc_${leg}_$c --pmc $c --output-format csv -d $O/pmc_${leg}_$c -- python3 $R/bench.py $(pmc_args_of $leg) || exit 1
    done
  done
fi
cd $R
find $O -name "*.csv" | wc -l
