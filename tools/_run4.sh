set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
for sc in 2 1.5; do timeout -k 10 300 python tools/time_config.py --config cfg5 --scale $sc --nt 40 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04d/t.log 2>&1; done
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 2 --nt 40 --rounds 2 --npml 16 >> gpurun_out/r04d/t.log 2>&1
cat gpurun_out/r04d/t.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "bf16 or fuzz or 8_row" > gpurun_out/r04d/t2.log 2>&1 || tail -30 gpurun_out/r04d/t2.log
tail -2 gpurun_out/r04d/t2.log
timeout -k 10 900 python tools/graph_probe.py --rounds 3 > gpurun_out/r04d/graph.jsonl 2> gpurun_out/r04d/graph.err || tail -20 gpurun_out/r04d/graph.err
cat gpurun_out/r04d/graph.jsonl
