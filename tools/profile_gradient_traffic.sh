# HBM-side traffic of the forward+store and adjoint+imaging kernels (256^3, 100 steps), separate PMC passes
set -o pipefail
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gt; mkdir -p $R/gpurun_out/gt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/gt/$c -- python3 $R/bench.py --mode gradient --nt 100 --steps 1 --warmup 0 > $R/gpurun_out/gt/$c.log 2>&1
done
find $R/gpurun_out/gt -name "*counter_collection.csv"
