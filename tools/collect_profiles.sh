set -o pipefail
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
python tests/parity_report.py gpurun_out/final/parity.json > gpurun_out/final/parity.log 2>&1; tail -3 gpurun_out/final/parity.log | cut -c1-200
python bench.py > gpurun_out/final/bench_forward.json 2> gpurun_out/final/bench_forward.err; tail -1 gpurun_out/final/bench_forward.json | cut -c1-200
python bench.py --mode gradient --steps 3 --warmup 1 > gpurun_out/final/bench_gradient.json 2>/dev/null
python bench.py --grid 512 --nt 100 --steps 3 --warmup 1 --cpu-steps 0 > gpurun_out/final/bench_512.json 2>/dev/null; tail -1 gpurun_out/final/bench_512.json | cut -c60-140
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt256 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/f256 -- python3 $R/bench.py --steps 1 --warmup 0 --nt 100 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/w256 -- python3 $R/bench.py --steps 1 --warmup 0 --nt 100 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt512 -- python3 $R/bench.py --grid 512 --nt 60 --steps 2 --warmup 1 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/f512 -- python3 $R/bench.py --grid 512 --nt 30 --steps 1 --warmup 0 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/w512 -- python3 $R/bench.py --grid 512 --nt 30 --steps 1 --warmup 0 --cpu-steps 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/ktgrad -- python3 $R/bench.py --mode gradient --steps 1 --warmup 1 > /dev/null 2>&1
cd $R; python tests/bench_mc.py --cpu-samples 100 > gpurun_out/final/mc.jsonl 2>/dev/null
ls gpurun_out/final/*/*/ | head -40
cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt2d -- python3 $R/tools/time_config.py --config cfg2 --nt 1000 --rounds 1 > $R/gpurun_out/final/time2d.log 2>&1
cd $R
