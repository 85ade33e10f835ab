# Round-4 end-to-end records of configs[2] (2-D 1024^2, 32 shots, 5 L-BFGS iterations) and configs[4] (3-D 256^3
# heterogeneous, 64 shots, 5 L-BFGS iterations) at FULL size on ONE GPU, on the final layout, with the sponge and with the
# convolutional PML; update form = shots.inversion_engine's choice (named in every record).
# Run on the GPU box: bash tools/collect_r04_runs.sh ; writes gpurun_out/r04_runs/*.json (copied to profiles/ by hand).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_runs
mkdir -p $O
cd $R
run() {  # name, args...
  local name=$1; shift
  echo "== $name: run_config.py $*"
  timeout -k 10 900 python3 tools/run_config.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; return 1; }
  cut -c1-400 $O/$name.json
}
run r04_run_cfg3_32shots_5lbfgs_1gpu_sponge --config cfg3 --iters 5 || exit 1
run r04_run_cfg3_32shots_5lbfgs_1gpu_cpml --config cfg3 --iters 5 --abc cpml || exit 1
run r04_run_cfg5_64shots_5lbfgs_1gpu_sponge --config cfg5 --iters 5 || exit 1
run r04_run_cfg5_64shots_5lbfgs_1gpu_cpml --config cfg5 --iters 5 --abc cpml || exit 1
