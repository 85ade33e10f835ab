# Round-3 tree against the round-4 tree on ONE box in ONE gpurun call (profiles/r04_ab_r03_vs_r04.json).
# Set-up (in the container, before the call; variant_build/ is git-ignored but travels to the GPU box):
#   git worktree add -f variant_build/r03tree cc82a3c && make -C variant_build/r03tree/full_waveform_inversion_amd/csrc -j8 \
#     && make -C variant_build/r03tree/oracle      (and `git worktree remove --force variant_build/r03tree` afterwards)
# Run on the GPU box: bash tools/ab_r03_vs_r04.sh
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ab
rm -f gpurun_out/r04ab/*
OLD=$GRAFT_REPO_ROOT/variant_build/r03tree
for rep in 1 2; do
for tree in $OLD $GRAFT_REPO_ROOT; do
  tag=$( [ $tree = $OLD ] && echo r03 || echo r04 )
  cd $tree
  for leg in fp64 cpml3d gradient gradient_increment cfg2_cpml; do
    timeout -k 10 300 python bench.py --leg $leg > $GRAFT_REPO_ROOT/gpurun_out/r04ab/${tag}_${leg}_$rep.json 2>/dev/null || echo "fail $tag $leg"
  done
  timeout -k 10 300 python bench.py --leg headline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r04ab/${tag}_headline_$rep.json 2>/dev/null || echo "fail headline"
  for sc in 1 1.5 2; do
    nt=$( [ $sc = 1 ] && echo 200 || echo 40 )
    timeout -k 10 300 python tools/time_config.py --config cfg5 --scale $sc --nt $nt --rounds 2 --abc cpml --npml 16 > $GRAFT_REPO_ROOT/gpurun_out/r04ab/${tag}_tc_${sc}_$rep.txt 2>&1 || echo "fail tc"
  done
done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04ab/*.json')):
    try:
        b = json.load(open(f))
    except Exception as ex:
        print(f, "unreadable"); continue
    if "legs" in b:
        for k, v in b["legs"].items():
            print(os.path.basename(f), k, v.get("kernel_avg_us"), v.get("us_per_time_step"), v.get("ms_per_shot_gradient"))
    elif "roofline" in b:
        print(os.path.basename(f), "headline", b["roofline"]["kernel_avg_us"], b["value"])
for f in sorted(glob.glob('gpurun_out/r04ab/*.txt')):
    print(os.path.basename(f), [l.split()[1] for l in open(f) if l.strip().startswith(("forward","save","adjoint"))])
PY
