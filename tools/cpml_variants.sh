# A/B of the 3-D CPML decompositions at 256^3 / npml 16 (tools/time_config.py): which axes ride in the step kernel, which
# run as line launches (fwi_pml.hip, pml_line), which as slab launches.  Run on the GPU box: bash tools/cpml_variants.sh [nt]
NT=${1:-200}
run() { echo "== $1"; shift; env "$@" python3 tools/time_config.py --config cfg5 --abc cpml --nt $NT --rounds 2 | tail -3; }
run "default: x in the step kernel, z + y line launches (8-byte lanes; the lane-width / streaming-hint A-B of DESIGN s.4 was run on builds with those as hooks)"
run "x + z in the step kernel, y line launch" FWI_STREAM_ZPML=1
run "plain step kernel, x slabs, z + y lines" FWI_NO_STREAM_XPML=1
run "round-2 form: x + z in the step kernel, y slabs" FWI_STREAM_ZPML=1 FWI_NO_PML_LINES=1
run "all slabs" FWI_NO_STREAM_XPML=1 FWI_NO_PML_LINES=1
