#!/usr/bin/env python3
"""The GPU suite's two seeded fuzz tests with other seeds, more cases and larger grids (several tiles, several z
chunks): forward + adjoint + gradient of random small configurations and of random option combinations (CPML /
increment form / bf16 store / checkpointing / imaging stride / two-step kernel) against the oracles.  A script, not a
test: `python tools/fuzz_soak.py [seeds] [first_seed] [maxdim]` on a GPU box; stops at the first mismatch with its tag."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    maxdim = int(sys.argv[3]) if len(sys.argv) > 3 else 41
    t0 = time.perf_counter()
    for seed in range(first, first + nseeds):
        T.fuzz_small_configurations(seed, 40, maxdim)
        T.fuzz_option_combinations(100000 + seed, 48, os.environ.__setitem__, max(8, maxdim - 4))
        print("seed %d ok (%.0f s)" % (seed, time.perf_counter() - t0), flush=True)
    print("ok: %d seeds x 88 cases, grids up to %d per axis" % (nseeds, maxdim - 1))


if __name__ == "__main__":
    main()
