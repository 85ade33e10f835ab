set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
for ty in 8 4; do
  for sc in 2 1.5; do
    echo "== TY=$ty scale=$sc" >> gpurun_out/r04a/ty.log
    FWI_STREAM_TY=$ty timeout -k 10 300 python tools/time_config.py --config cfg5 --scale $sc --nt 40 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04a/ty.log 2>&1
  done
done
echo "== 256 default" >> gpurun_out/r04a/ty.log
timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 1 --nt 200 --rounds 2 --abc cpml --npml 16 >> gpurun_out/r04a/ty.log 2>&1
echo "== 512 sponge TY8 / TY4" >> gpurun_out/r04a/ty.log
for ty in 8 4; do FWI_STREAM_TY=$ty timeout -k 10 300 python tools/time_config.py --config cfg5 --scale 2 --nt 40 --rounds 2 --npml 16 >> gpurun_out/r04a/ty.log 2>&1; done
cat gpurun_out/r04a/ty.log
