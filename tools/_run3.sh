set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r04c/gpu_suite.log 2>&1 || { tail -60 gpurun_out/r04c/gpu_suite.log; echo SUITE_FAILED; }
tail -3 gpurun_out/r04c/gpu_suite.log
for sc in 2 1.5; do timeout -k 10 300 python tools/time_config.py --config cfg5 --scale $sc --nt 40 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04c/t.log 2>&1; done
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04c/t.log 2>&1
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 --update-form increment >> gpurun_out/r04c/t.log 2>&1
cat gpurun_out/r04c/t.log
