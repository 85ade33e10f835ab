set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04/gpu_suite3.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite3.log; echo SUITE_FAILED; }
tail -2 gpurun_out/r04/gpu_suite3.log
timeout -k 10 1100 python tests/parity_report.py gpurun_out/r04/r04_parity.json > gpurun_out/r04/parity.log 2>&1 || { tail -20 gpurun_out/r04/parity.log; echo PARITY_FAILED; }
python - <<'PY'
import json
for r in json.load(open('gpurun_out/r04/r04_parity.json')):
    print("%-58s seis %.1e adj %.1e g %.1e e2e %.1e" % (r["case"][:58], r["seis"], r["adj_src"], r["grad_same_r"], r["grad_e2e"]))
PY
python -c "import __graft_entry__ as g; g.smoke()"
