#!/bin/bash
out=gpurun_out/place5.log; : > $out
for rep in 1 2; do
for leg in cpml512 cpml3d_adjoint; do
for m in 0 1; do
echo "## $leg tune=$m" >> $out
if [ $m = 0 ]; then FWI_PLACEMENT_TUNE=0 python bench.py --leg $leg 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.readline()); r=b['legs']['$leg']; print(r['kernel_avg_us'], r['frac'])" >> $out || exit 1
else python bench.py --leg $leg 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.readline()); r=b['legs']['$leg']; print(r['kernel_avg_us'], r['frac'])" >> $out || exit 1; fi
done; done; done
