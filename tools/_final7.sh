set -e
cd $GRAFT_REPO_ROOT
LEGS="gradient_increment cpml3d_adjoint" timeout -k 10 900 bash tools/collect_r04.sh legs_only > gpurun_out/r04/collect3.log 2>&1 || true
