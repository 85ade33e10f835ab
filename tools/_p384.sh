set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_cpml.py -q -x > gpurun_out/r04/cpml_dpp.log 2>&1 || { tail -30 gpurun_out/r04/cpml_dpp.log; echo CPML_FAILED; exit 1; }
tail -1 gpurun_out/r04/cpml_dpp.log
for sc in 1 1.25 1.5 1.75 2; do nt=$( [ $sc = 1 ] && echo 200 || echo 40 ); python tools/time_config.py --config cfg5 --scale $sc --nt $nt --rounds 2 --abc cpml --npml 16 | grep -E "cfg5|forward|save|adjoint" | cut -c1-90; done
python bench.py --leg cpml3d | python -c "import json,sys; b=json.load(sys.stdin)['legs']['cpml3d']; print('cpml3d leg', b['us_per_time_step'], b['frac'])"
