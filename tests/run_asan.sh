#!/bin/bash
# Host sanitizer pass: AddressSanitizer + UBSan builds of the C-ABI shim (host side) and of the C
# oracle, then the CPU-only tests that exercise them.  GPU sanitizers are not available on this pool (xnack), so
# this covers the host code: argument validation, error paths, string handling, the oracle's index arithmetic.
#   tests/run_asan.sh [logfile]        (test infrastructure: it loads the oracle, so it lives under tests/)
# Exit status: non-zero when a build fails, a pytest run fails, or a sanitizer report appears in the output
# (the status used to be tee's, and reports scrolled by).
set -uo pipefail
cd "$(dirname "$0")/.."
LOG=${1:-/dev/stdout}
make -C full_waveform_inversion_amd/csrc -s asan || exit 1
make -C oracle -s asan || exit 1
CLANG_RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
GCC_RT=$(gcc -print-file-name=libasan.so)
TMP=$(mktemp)
trap 'rm -f "$TMP"' EXIT
rc=0
{
  echo "== shim (hipcc host code, $CLANG_RT) =="
  LD_PRELOAD=$CLANG_RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
    FWI_HIP_LIB=$PWD/full_waveform_inversion_amd/libfwi_hip_asan.so \
    python -m pytest tests/test_abi.py tests/test_c_client.py -q -p no:cacheprovider -m "not gpu" 2>&1
  echo "pytest-rc-shim=$?"
  echo "== C oracle (gcc, $GCC_RT) =="
  LD_PRELOAD=$GCC_RT ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=2 \
    FWI_ORACLE_LIB=$PWD/oracle/libfwi_oracle_asan.so \
    python -m pytest tests/test_oracle.py -q -p no:cacheprovider -k "c_ or C or port or cpml" 2>&1
  echo "pytest-rc-oracle=$?"
} > "$TMP"
grep -q "pytest-rc-shim=0" "$TMP" || rc=1
grep -q "pytest-rc-oracle=0" "$TMP" || rc=1
if grep -q -E "ERROR: AddressSanitizer|runtime error:|ERROR: LeakSanitizer" "$TMP"; then rc=1; fi
# the log keeps the sanitizer reports in full and the last lines of each pytest run
{ grep -E "^== |pytest-rc-|ERROR: AddressSanitizer|runtime error:|passed|failed|error" "$TMP"; echo "run_asan rc=$rc"; } | tee "$LOG"
exit $rc
