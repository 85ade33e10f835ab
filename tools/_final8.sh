set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 700 python tools/fuzz_soak.py 2 601 299 > gpurun_out/r04/soak299.log 2>&1 || { tail -20 gpurun_out/r04/soak299.log; echo SOAK_FAILED; }
tail -2 gpurun_out/r04/soak299.log
timeout -k 10 400 python tools/size_sweep.py --ndim 3 --sizes 128,192,256,320,384,512,640 --abc cpml --npml 16 > gpurun_out/r04/size_sweep_cpml.jsonl 2>&1 || tail -5 gpurun_out/r04/size_sweep_cpml.jsonl
cat gpurun_out/r04/size_sweep_cpml.jsonl | cut -c1-200
