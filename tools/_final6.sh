set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04/gpu_suite2.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite2.log; echo SUITE_FAILED; }
tail -2 gpurun_out/r04/gpu_suite2.log
for i in 1 2; do timeout -k 10 300 python bench.py --leg gradient_increment > gpurun_out/r04/ginc_$i.json 2>/dev/null; python -c "
import json; b=json.load(open('gpurun_out/r04/ginc_$i.json'))['legs']['gradient_increment']; print(b['kernel_avg_us'], b['ms_per_shot_gradient'], b['frac'])"; done
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 --update-form increment
timeout -k 10 600 python tools/fuzz_soak.py 6 501 97 > gpurun_out/r04/soak97.log 2>&1 || { tail -20 gpurun_out/r04/soak97.log; echo SOAK_FAILED; }
tail -2 gpurun_out/r04/soak97.log
