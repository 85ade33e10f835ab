# A/B of the 3-D CPML decompositions at 256^3 / npml 16 (tools/time_config.py): which axes ride in the step kernel, which
# hand their term over from the line launch (fwi_pml.hip, pml_line_t), which run as slab launches around the step kernel.
# Run on the GPU box: bash tools/cpml_variants.sh [nt]
NT=${1:-200}
run() { echo "== $1"; shift; env "$@" python3 tools/time_config.py --config cfg5 --abc cpml --nt $NT --rounds 2 | tail -3; }
run "default: ONE line launch (z + y, 8-byte lanes) hands the term over, x in the step kernel's lanes"
run "the same with 16-byte lanes in the line launch" FWI_PML_LINE_VL=4
run "plain step kernel + the handed-over z / y terms, x slabs" FWI_NO_STREAM_XPML=1
run "all slabs (round-2 form)" FWI_NO_PML_LINES=1
