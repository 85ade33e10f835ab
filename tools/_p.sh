#!/bin/bash
out=gpurun_out/place6.log; : > $out
P="python tools/variance_probe.py --contexts 2 --shots 2 --abc sponge --update-form increment"
for rep in 1 2 3 4 5; do
echo "## increment sponge, search" >> $out; $P 2>&1 | grep context >> $out || exit 1
echo "## increment sponge, search + 3 redraws" >> $out; FWI_PLACEMENT_REDRAW=3 $P 2>&1 | grep context >> $out || exit 1
done
