set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout -k 10 900 python -m pytest tests/test_gpu_cpml.py -x -q > gpurun_out/r04b/cpml.log 2>&1 || { tail -40 gpurun_out/r04b/cpml.log; exit 1; }
tail -3 gpurun_out/r04b/cpml.log
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 > gpurun_out/r04b/t256.log 2>&1
cat gpurun_out/r04b/t256.log
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 2 --nt 40 --rounds 2 --abc cpml --npml 16 > gpurun_out/r04b/t512.log 2>&1
cat gpurun_out/r04b/t512.log
