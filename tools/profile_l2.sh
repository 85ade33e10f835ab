set -o pipefail
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/l2; mkdir -p $R/gpurun_out/l2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/l2/p512 -- python3 $R/bench.py --grid 512 --nt 30 --steps 1 --warmup 0 --cpu-steps 0 > $R/gpurun_out/l2/p512.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/l2/p256 -- python3 $R/bench.py --nt 100 --steps 1 --warmup 0 --cpu-steps 0 > $R/gpurun_out/l2/p256.log 2>&1
find $R/gpurun_out/l2 -name "*counter_collection.csv" | head
