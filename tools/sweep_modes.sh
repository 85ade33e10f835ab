mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest5.log; tail -3 gpurun_out/pytest5.log
for m in save adjoint; do timeout -k 10 150 python tools/tune_stream.py --nt 100 --rounds 2 --ty 4,8 --zchunk 16,32,64 --point 0 --mode $m > gpurun_out/tune_${m}2.log 2>&1; tail -8 gpurun_out/tune_${m}2.log; done
timeout -k 10 150 python tools/tune_stream.py --nt 400 --rounds 2 --ty 4,8 --zchunk 16,32,64 --npml 16 --point 0 > gpurun_out/tune_damp2.log 2>&1; tail -8 gpurun_out/tune_damp2.log
