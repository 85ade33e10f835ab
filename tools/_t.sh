cd $GRAFT_REPO_ROOT
for sc in 1 1.25 1.5 2; do nt=$( [ $sc = 1 ] && echo 200 || echo 40 ); echo "cpml scale $sc"; FWI_DEBUG_PML=1 python tools/time_config.py --config cfg5 --scale $sc --nt $nt --rounds 2 --abc cpml --npml 16 2>&1 | grep -E "fwi:|forward|save|adjoint" | cut -c1-100; done
echo "256 increment"; python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --npml 16 --update-form increment | grep -E "forward|save|adjoint"
echo "256 cpml increment"; python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --npml 16 --update-form increment --abc cpml | grep -E "forward|save|adjoint"
echo "512 increment"; python tools/time_config.py --config cfg5 --scale 2 --nt 40 --rounds 2 --npml 16 --update-form increment | grep -E "forward|save|adjoint"
for i in 1 2; do python bench.py --leg cpml3d | python -c "import json,sys; b=json.load(sys.stdin)['legs']['cpml3d']; print('cpml3d leg', b['us_per_time_step'], b['frac'])"; done
python bench.py --leg gradient_increment | python -c "import json,sys; b=json.load(sys.stdin)['legs']['gradient_increment']; print('ginc leg', b['kernel_avg_us'], b['ms_per_shot_gradient'], b['frac'])"
python bench.py --leg headline --steps 3 --warmup 1 | python -c "import json,sys; b=json.load(sys.stdin); print('headline', b['value'], b['roofline']['kernel_avg_us'])"
timeout -k 10 600 python -m pytest tests/test_gpu_cpml.py tests/test_gpu_round4.py -q -x 2>&1 | tail -2
